"""Parity under weight-distribution stress (round-5 verdict, item 3).

Every other parity test runs on ONE benign draw per model (synth_inputs: seed 1234 / 4321 / 987, He-scaled convs, damped
residual gains), whose fp32 activations peak at ~70 (Wav2Lip) and ~14 (MuseTalk): nowhere near the fp16 range the device
computes in.  The reference loads real checkpoints (avatars/wav2lip_avatar.py:51-70, avatars/musetalk/utils/utils.py:15-31),
whose activations nobody here has seen.  These tests walk a family of draws - three seeds per model x BatchNorm / GroupNorm gain
multipliers that push the fp32 peak activation through 1e3 and 1e4 to past the fp16 limit (synth_inputs.wav2lip_state_dict(gain=),
musetalk_unet_state_dict(gn_gain=), vae_decoder_state_dict(gn_gain=)) - and at every point:

  * run the device with knob SAT_CHECK on and read the saturation counters (ltk_debug_saturation: values an epilogue clamped to
    the fp16 / e4m3 limit, non-finite values);
  * where both counters are 0, hold the product frames (knob off) to the stated tolerance against the fp32 oracle:
    PSNR >= 40 dB and max-abs <= 6 LSB (fp16 paths); fp8 conv path: worst frame >= 36.5 dB - what the three draws support
    (measured 37.0 / 37.7 dB here on seeds 23 / 11 and 39.2 dB on the draw tests/test_fp8_gpu.py gates at 38 dB pooled / 37 dB
    worst frame: round 6 found that gate to be a property of that one draw and narrowed the claim);
  * where they are not, the device is no longer computing the reference's numbers and SAYS so: the counter is the loud failure
    (asserted non-zero at the points built to overflow), and the tolerance is not claimed there.

The table these tests print is in DESIGN.md section 4.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import synth_inputs as synth  # noqa: E402


def _psnr_lsb(got, ref):
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    worst = 99.0
    for i in range(got.shape[0]):
        mse = float((d[i].astype(np.float64) ** 2).mean())
        worst = min(worst, 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse))
    return worst, int(d.max())


# (seed, gain, expectation): "ok" = counters must be 0 and the tolerance must hold; "sat" = built to overflow fp16: counters must fire
W2L_POINTS = [(1234, 1, "ok"), (1234, 16, "ok"), (1234, 256, "ok"),
              (7, 1, "ok"), (7, 16, "ok"), (7, 256, "ok"),
              (99, 1, "ok"), (99, 16, "ok"), (99, 256, "ok"),
              (1234, 4096, "sat")]


@pytest.mark.gpu
def test_wav2lip_frames_vs_oracle_over_seeds_and_gains():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    from oracle import mel_oracle, plugin_oracle, wav2lip_oracle as W
    B = 2
    frames, faces, coords = synth.wav2lip_avatar(n_frames=3, full_hw=(360, 640), box=160, seed=0)
    audio = synth.synthetic_audio(2.0)
    n_chunks = 20 + 2 * B
    ref_mel = np.stack(mel_oracle.mel_chunks(audio[: n_chunks * 320], n_chunks))[:B]
    rows = []
    for seed, gain, expect in W2L_POINTS:
        sd_np = synth.wav2lip_state_dict(seed, float(gain))
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
        taps = {}
        with torch.no_grad():
            mel_t, img_t = plugin_oracle.pack_inputs(faces, 1, B, list(ref_mel))
            pred = W.forward(sd, mel_t, img_t, taps)
        ref = (pred.numpy().transpose(0, 2, 3, 1) * 255.).astype(np.uint8)
        peak = max(float(t.abs().max()) for t in taps.values())
        eng = Engine(0)
        try:
            eng.load_wav2lip(sd_np, max_frames=4)
            aid = eng.register_avatar(faces, frames, coords)
            d_mel = torch.from_numpy(ref_mel.astype(np.float32)).cuda()
            d_pred = torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device="cuda")
            Engine.set_knob("SAT_CHECK", 1)
            try:
                eng.saturation(reset=True)
                eng.wav2lip_infer([(aid, 1, B, d_mel.data_ptr(), d_pred.data_ptr())])
                at_limit, nonfinite = eng.saturation(reset=True)
            finally:
                Engine.set_knob("SAT_CHECK", 0)
            d_pred.zero_()
            eng.wav2lip_infer([(aid, 1, B, d_mel.data_ptr(), d_pred.data_ptr())])      # the product path (fused head)
            psnr, lsb = _psnr_lsb(d_pred.cpu().numpy(), ref)
        finally:
            eng.close()
        rows.append((seed, gain, peak, at_limit, nonfinite, psnr, lsb, expect))
        print(f"[stress w2l] seed {seed:5d} gain {gain:5d}: fp32 peak |act| {peak:10.1f}  at-limit {at_limit:8d}  non-finite {nonfinite:6d}  "
              f"worst-frame PSNR {psnr:6.2f} dB  max {lsb:3d} LSB  ({expect})", flush=True)
    for seed, gain, peak, at_limit, nonfinite, psnr, lsb, expect in rows:
        if expect == "ok":
            assert at_limit == 0 and nonfinite == 0, f"seed {seed} gain {gain}: unexpected saturation ({at_limit}, {nonfinite}) at fp32 peak {peak:.0f}"
            assert psnr >= 40.0 and lsb <= 6, f"seed {seed} gain {gain}: {psnr:.2f} dB / {lsb} LSB with clean counters"
        else:
            assert peak > 65504.0, "the overflow point does not overflow any more: raise its gain"
            assert at_limit + nonfinite > 0, f"seed {seed} gain {gain}: fp32 peak {peak:.0f} is past the fp16 limit and the counters stayed silent"


# (unet/vae seed pair, GroupNorm gain, frames, fp8, expectation)
MT_POINTS = [((4321, 987), 1, 1, False, "ok"), ((11, 12), 1, 3, False, "ok"), ((23, 24), 1, 8, False, "ok"),
             ((4321, 987), 256, 1, False, "ok"), ((4321, 987), 2048, 1, False, "ok"),
             ((4321, 987), 16384, 1, False, "sat"),
             ((11, 12), 1, 1, True, "ok"), ((23, 24), 1, 1, True, "ok"),
             ((4321, 987), 64, 1, True, "sat")]


@pytest.mark.gpu
def test_musetalk_frames_vs_oracle_over_seeds_gains_and_batch_sizes():
    """MuseTalk end to end (PE + U-Net + VAE decode + uint8) against the fp32 oracle: three seeds, at B = 1, 3 and 8 (the tile /
    split rules of csrc/musetalk.hip follow the frame count; the other tests run B = 2, 16, 64), GroupNorm gains up to a fp32
    peak of 1.5e4, the overflow point, and the fp8 conv path on the other two seeds and at its own saturation point (e4m3
    saturates at |GroupNorm-SiLU output| x 8 = 448)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    from oracle import musetalk_oracle as M
    rows = []
    for (useed, vseed), gain, Bf, fp8, expect in MT_POINTS:
        unet_sd = synth.musetalk_unet_state_dict(useed, gn_gain=float(gain))
        vae_sd = synth.vae_decoder_state_dict(vseed, gn_gain=float(gain))
        lat = np.concatenate(synth.musetalk_latents(Bf, seed=5 + useed % 7))
        feat = synth.musetalk_whisper_feats(Bf, seed=11 + useed % 5)
        usd = {k: torch.from_numpy(v) for k, v in unet_sd.items()}
        vsd = {k: torch.from_numpy(v) for k, v in vae_sd.items()}
        taps, vtaps = {}, {}
        with torch.no_grad():
            ref_lat = M.unet_forward(usd, torch.from_numpy(lat), M.positional_encoding(torch.from_numpy(feat)), taps=taps, detail="")
            ref_img = M.vae_decode(vsd, ref_lat / M.VAE_SCALING, vtaps)
        ref = ((ref_img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).round().astype(np.uint8)[..., ::-1]
        peak = max(max(float(t.abs().max()) for t in taps.values()), max(float(t.abs().max()) for t in vtaps.values()))
        eng = Engine(0)
        try:
            eng.load_musetalk(unet_sd, vae_sd, max_frames=Bf, fp8=fp8)
            Engine.set_knob("SAT_CHECK", 1)
            try:
                eng.saturation(reset=True)
                eng.musetalk_forward_host(lat, feat)
                at_limit, nonfinite = eng.saturation(reset=True)
            finally:
                Engine.set_knob("SAT_CHECK", 0)
            _, _, got = eng.musetalk_forward_host(lat, feat)
            psnr, lsb = _psnr_lsb(got, ref)
        finally:
            eng.close()
        rows.append((useed, gain, Bf, fp8, peak, at_limit, nonfinite, psnr, lsb, expect))
        print(f"[stress mt] seed {useed:5d} gn-gain {gain:6d} B={Bf} {'fp8 ' if fp8 else 'fp16'}: fp32 peak |act| {peak:10.1f}  at-limit {at_limit:9d}  "
              f"non-finite {nonfinite:8d}  worst-frame PSNR {psnr:6.2f} dB  max {lsb:3d} LSB  ({expect})", flush=True)
    for useed, gain, Bf, fp8, peak, at_limit, nonfinite, psnr, lsb, expect in rows:
        tag = f"seed {useed} gn-gain {gain} B={Bf} {'fp8' if fp8 else 'fp16'}"
        if expect == "ok":
            assert at_limit == 0 and nonfinite == 0, f"{tag}: unexpected saturation ({at_limit}, {nonfinite}) at fp32 peak {peak:.0f}"
            if fp8:
                assert psnr >= 36.5, f"{tag}: {psnr:.2f} dB with clean counters (fp8 gate over weight draws: 36.5 dB worst frame)"
            else:
                assert psnr >= 40.0 and lsb <= 6, f"{tag}: {psnr:.2f} dB / {lsb} LSB with clean counters"
        else:
            assert at_limit + nonfinite > 0, f"{tag}: built to saturate (fp32 peak {peak:.0f}) and the counters stayed silent"
