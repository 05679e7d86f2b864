"""Parity of the HIP MuseTalk path (U-Net + VAE decoder, csrc/musetalk.hip + nn_kernels.hip) against the oracle
(oracle/musetalk_oracle.py: a restatement of the diffusers graph - PARITY UNPINNED against diffusers itself, see
its header) on seeded weights and inputs.

Tolerances (fp16 activations / fp32 accumulate on the device vs the fp32 oracle; the reference itself runs this
path in fp16, avatars/musetalk_avatar.py:62-64):
  per tensor : relative L2 <= 1e-2 (measured ~1.5e-3)
  latents    : relative L2 <= 1e-2 on the U-Net output
  frames     : PSNR >= 45 dB and >= 99.9 % of the bytes within +-2 LSB of the oracle decode_latents (measured 60.5 dB, max 1 LSB)
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import synth_inputs as synth  # noqa: E402
from oracle import musetalk_oracle as M  # noqa: E402

B = 2


@pytest.fixture(scope="module")
def mt():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    unet_sd = synth.musetalk_unet_state_dict()
    vae_sd = synth.vae_decoder_state_dict()
    eng = Engine(0)
    eng.load_musetalk(unet_sd, vae_sd, max_frames=B)
    yield eng, {k: torch.from_numpy(v) for k, v in unet_sd.items()}, {k: torch.from_numpy(v) for k, v in vae_sd.items()}
    eng.close()


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-9))


# Intermediates the default program no longer writes to memory (knob MT_FUSE, csrc/tune.h): the GEGLU projection lives in the
# accumulators of its own epilogue (bit 0), the LayerNorm outputs in the epilogues of the projections that consume them (bit 2).
# `test_unfused_program_every_tap_vs_oracle` builds the program with the fusions off and checks these taps too.
FUSED_AWAY = (".ff.net.0.proj", "transformer_blocks.0.norm1", "transformer_blocks.0.norm2", "transformer_blocks.0.norm3")
MT_FUSE_DEFAULT = 7          # csrc/tune.hip


def _check_unet_taps(eng, usd, fused):
    lat = np.concatenate(synth.musetalk_latents(B))
    feat = synth.musetalk_whisper_feats(B)
    taps = {}
    with torch.no_grad():
        pe = M.positional_encoding(torch.from_numpy(feat))
        ref_lat = M.unet_forward(usd, torch.from_numpy(lat), pe, taps=taps, detail="down_blocks.0")
    got_lat, _, _ = eng.musetalk_forward_host(lat, feat)
    report, seen = [], 0
    for name, t in taps.items():
        if fused and name.endswith(FUSED_AWAY):
            continue
        ref = t.numpy()
        shape = list(ref.shape)
        if name.endswith(".attn"):
            d = shape[1] // 8
            d16 = (d + 15) // 16 * 16
            dev = eng.musetalk_debug_get(name, (shape[0], 8 * d16, shape[2], shape[3]))
            dev = dev.reshape(shape[0], 8, d16, shape[2], shape[3])[:, :, :d].reshape(shape)
        else:
            c16 = (shape[1] + 15) // 16 * 16
            dev = eng.musetalk_debug_get(name, (shape[0], c16, shape[2], shape[3]))[:, :shape[1]]
        r = rel_l2(dev, ref)
        seen += 1
        if not (r <= 1e-2):
            report.append(f"{name}: rel L2 {r:.3e}")
    r = rel_l2(got_lat, ref_lat.numpy())
    print(f"[mt] {'fused' if fused else 'unfused'} program: {seen} taps, {len(eng.musetalk_ops())} ops, U-Net output rel_l2={r:.3e}")
    assert r <= 1e-2 and not report, "\n".join(report)
    return got_lat


@pytest.mark.gpu
def test_unfused_program_every_tap_vs_oracle(mt):
    """The program built with every round-6 fusion off (LTK_MT_FUSE=0, LTK_MT_GN1=0: the round-5 launch list) holds every
    op-level tensor of the first down block: all of them against the oracle; and the default program's U-Net output equals the
    unfused program's within fp16 rounding of the intermediates it no longer rounds."""
    from livetalking_amd.engine import Engine
    eng0, usd, vsd = mt
    got_fused = _check_unet_taps(eng0, usd, True)
    Engine.set_knob("MT_FUSE", 0)
    Engine.set_knob("MT_GN1", 0)
    try:
        eng = Engine(0)
        eng.load_musetalk({k: v.numpy() for k, v in usd.items()}, {k: v.numpy() for k, v in vsd.items()}, max_frames=B)
        got_plain = _check_unet_taps(eng, usd, False)
        eng.close()
    finally:
        Engine.set_knob("MT_FUSE", MT_FUSE_DEFAULT)
        Engine.set_knob("MT_GN1", 1)
    r = rel_l2(got_fused, got_plain)
    print(f"[mt] fused vs unfused program, U-Net output rel_l2={r:.3e}")
    assert r <= 3e-3


@pytest.mark.gpu
def test_unet_and_vae_vs_oracle(mt):
    eng, usd, vsd = mt
    lat = np.concatenate(synth.musetalk_latents(B))
    feat = synth.musetalk_whisper_feats(B)
    taps = {}
    with torch.no_grad():
        pe = M.positional_encoding(torch.from_numpy(feat))
        ref_lat = M.unet_forward(usd, torch.from_numpy(lat), pe, taps=taps, detail="down_blocks.0")
        vtaps = {}
        ref_img = M.vae_decode(vsd, ref_lat / M.VAE_SCALING, vtaps)
    got_lat, got_img, got_frames = eng.musetalk_forward_host(lat, feat)

    report = []

    def check(name, ref, tol=1e-2, heads=None):
        ref = ref.numpy()
        shape = list(ref.shape)
        if heads:      # padded heads on the device
            d = shape[1] // heads
            d16 = (d + 15) // 16 * 16
            dev = eng.musetalk_debug_get(name, (shape[0], heads * d16, shape[2], shape[3]))
            dev = dev.reshape(shape[0], heads, d16, shape[2], shape[3])[:, :, :d].reshape(shape)
        else:
            c16 = (shape[1] + 15) // 16 * 16
            dev = eng.musetalk_debug_get(name, (shape[0], c16, shape[2], shape[3]))[:, :shape[1]]
        r = rel_l2(dev, ref)
        print(f"[mt] {name:70s} rel_l2={r:.3e} refmax={np.abs(ref).max():.3g}")
        if not (r <= tol):
            report.append(f"{name}: rel L2 {r:.3e}")

    # op-level taps of the first down block (every kernel type appears there), then block-level taps
    for name, t in taps.items():
        if name.count(".") >= 3 and not name.endswith(FUSED_AWAY):
            check(name, t, heads=8 if name.endswith(".attn") else None)
    for name, t in taps.items():
        if name.count(".") < 3:
            check(name, t)
    r = rel_l2(got_lat, ref_lat.numpy())
    print(f"[mt] U-Net output rel_l2={r:.3e}")
    assert r <= 1e-2, report
    assert not report, "\n".join(report)

    # VAE decoder, fed with the DEVICE's latents so that its error is measured on its own
    with torch.no_grad():
        vt2 = {}
        ref_img2 = M.vae_decode(vsd, torch.from_numpy(got_lat) / M.VAE_SCALING, vt2)
    # the device decoded its own fp16 latents; got_lat is their fp32 copy
    for name, t in vt2.items():
        shape = list(t.shape)
        c16 = (shape[1] + 15) // 16 * 16
        dev = eng.musetalk_debug_get(name, (shape[0], c16, shape[2], shape[3]))[:, :shape[1]]
        rr = rel_l2(dev, t.numpy())
        print(f"[mt] {name:70s} rel_l2={rr:.3e}")
        if not (rr <= 1e-2):
            report.append(f"{name}: rel L2 {rr:.3e}")
    assert not report, "\n".join(report)
    ri = rel_l2(got_img, ref_img2.numpy())
    print(f"[mt] VAE image rel_l2={ri:.3e} (vs oracle on the same latents); end-to-end vs oracle {rel_l2(got_img, ref_img.numpy()):.3e}")
    ref_frames = ((ref_img2 / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).round().astype(np.uint8)[..., ::-1]
    d = np.abs(got_frames.astype(np.int32) - ref_frames.astype(np.int32))
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    print(f"[mt] frames PSNR {psnr:.2f} dB, max {d.max()} LSB, within+-2: {float((d <= 2).mean()):.5f}")
    assert psnr >= 45.0 and float((d <= 2).mean()) >= 0.999
    # END TO END: PE + U-Net + VAE on the device against the oracle decoding the ORACLE's own latents (no device value enters
    # the reference side): frames PSNR >= 40 dB, max-abs <= 6 LSB, >= 99 % of the bytes within +-2 LSB (the Wav2Lip frame tolerance)
    e2e_frames = ((ref_img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).round().astype(np.uint8)[..., ::-1]
    de = np.abs(got_frames.astype(np.int32) - e2e_frames.astype(np.int32))
    mse_e = float((de.astype(np.float64) ** 2).mean())
    psnr_e = 99.0 if mse_e == 0 else 10 * np.log10(255.0 ** 2 / mse_e)
    print(f"[mt] END-TO-END frames PSNR {psnr_e:.2f} dB, max {de.max()} LSB, within+-2: {float((de <= 2).mean()):.5f}")
    assert psnr_e >= 40.0 and de.max() <= 6 and float((de <= 2).mean()) >= 0.99


@pytest.mark.gpu
def test_musetalk_infer_and_blend(mt):
    """ltk_musetalk_infer (latent gather by mirror_index + PE + U-Net + VAE + uint8 BGR) equals the host-input hook on
    the same frames, and ltk_paste_blend matches the oracle's paste_back_frame bit for bit."""
    from oracle import paste_oracle
    eng, usd, vsd = mt
    n = 3
    lats = synth.musetalk_latents(n)
    frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(360, 640), box=160, seed=2)
    face_boxes, crop_boxes, masks = [], [], []
    rng = np.random.default_rng(3)
    for i in range(n):
        x1, y1 = 240 + int(rng.integers(-3, 4)), 100 + int(rng.integers(-3, 4))
        x2, y2 = x1 + 150 + int(rng.integers(0, 9)), y1 + 170 + int(rng.integers(0, 9))
        xs, ys, xe, ye = x1 - 30, y1 - 40, x2 + 30, y2 + 35
        face_boxes.append((x1, y1, x2, y2)); crop_boxes.append((xs, ys, xe, ye))
        m = np.zeros((ye - ys, xe - xs), np.float64)
        m[(ye - ys) // 2:, 20:-20] = 255.0
        # soft edge, like the Gaussian-blurred masks of avatars/musetalk/utils/blending.py:129-135
        k = np.ones(15) / 15
        m = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, m)
        m = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, m)
        m8 = np.clip(np.rint(m), 0, 255).astype(np.uint8)
        masks.append(np.repeat(m8[:, :, None], 3, axis=2))
    aid = eng.register_musetalk_avatar(lats, frames, face_boxes, masks, crop_boxes)
    feat = synth.musetalk_whisper_feats(B, seed=21)
    d_feat = torch.from_numpy(feat).cuda()
    d_pred = torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device="cuda")
    index = 2          # frames mirror_index(3, 2) = 2, mirror_index(3, 3) = 2
    eng.musetalk_infer([(aid, index, B, d_feat.data_ptr(), d_pred.data_ptr())])
    idxs = [paste_oracle.mirror_index(n, index + i) for i in range(B)]
    _, _, want = eng.musetalk_forward_host(np.concatenate([lats[i] for i in idxs]), feat)
    got = d_pred.cpu().numpy()
    assert np.array_equal(got, want), f"infer vs host hook: max diff {np.abs(got.astype(int) - want.astype(int)).max()}"
    out = np.empty((360, 640, 3), np.uint8)
    eng.paste_blend(aid, idxs[0], d_pred[0].data_ptr(), out)
    ref = paste_oracle.paste_blend_frame(got[0], frames[idxs[0]], face_boxes[idxs[0]], masks[idxs[0]], crop_boxes[idxs[0]])
    dd = np.abs(out.astype(int) - ref.astype(int))
    print(f"[mt] paste_blend max diff {dd.max()}, differing bytes {(dd != 0).sum()}")
    assert np.array_equal(out, ref)


@pytest.mark.gpu
def test_vae_encoder_avatar_prep():
    """ltk_vae_encode_faces (avatar preparation, vae.get_latents_for_unet) vs the oracle: the distribution mean and a
    sample with explicit noise, rel. L2 <= 1e-2 (fp16 activations; the reference runs this in fp16 too,
    avatars/musetalk/genavatar.py:108)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    sd_np = synth.vae_encoder_state_dict()
    eng = Engine(0)
    eng.load_vae_encoder(sd_np, max_faces=2)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    _, faces, _ = synth.wav2lip_avatar(n_frames=2, full_hw=(64, 64), box=16, seed=9)
    noise = np.random.default_rng(1).standard_normal((2, 2, 4, 32, 32)).astype(np.float32)
    got_mean = eng.vae_encode_faces(np.stack(faces))
    got_samp = eng.vae_encode_faces(np.stack(faces), noise)
    with torch.no_grad():
        ref_mean = torch.cat([M.get_latents_for_unet(sd, f) for f in faces]).numpy()
        ref_samp = torch.cat([M.get_latents_for_unet(sd, f, torch.from_numpy(noise[i])) for i, f in enumerate(faces)]).numpy()
    r1, r2 = rel_l2(got_mean, ref_mean), rel_l2(got_samp, ref_samp)
    print(f"[mt] vae encoder: mean rel_l2={r1:.3e}, sample rel_l2={r2:.3e}, |latent| max {np.abs(ref_mean).max():.3g}")
    assert got_mean.shape == (2, 8, 32, 32) and r1 <= 1e-2 and r2 <= 1e-2
    eng.close()


@pytest.mark.gpu
def test_full_size_batching_properties():
    """BASELINE.json configs[4]'s per-GPU share at full size: 4 sessions x 16 frames in ONE call (64 frames) against the same
    sessions one by one.  A session's frames do not depend on what it was batched with, up to the summation order of the
    per-launch split-K slabs and GroupNorm segments (both follow the frame count): <= 2 LSB, PSNR >= 50 dB; two sessions with the
    same latents and audio get the same bytes inside one call; a repeated call is bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    S, Bf, n = 4, 16, 5
    eng = Engine(0)
    try:
        eng.load_musetalk(synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict(), max_frames=S * Bf)
        lats = synth.musetalk_latents(n)
        frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(360, 640), box=160, seed=2)
        boxes = [(240, 100, 400, 280)] * n
        crops = [(210, 60, 430, 315)] * n
        masks = [np.full((255, 220, 3), 255, np.uint8)] * n
        aid = eng.register_musetalk_avatar(lats, frames, boxes, masks, crops)
        feats = [torch.from_numpy(synth.musetalk_whisper_feats(Bf, seed=30 + s)).cuda() for s in range(S)]
        feats[3] = feats[1]
        index = [0, 3, 7, 3]                                  # sessions 1 and 3: same latents, same audio
        single = torch.zeros(S, Bf, 256, 256, 3, dtype=torch.uint8, device="cuda")
        for s in range(S):
            eng.musetalk_infer([(aid, index[s], Bf, feats[s].data_ptr(), single[s].data_ptr())])
        assert torch.equal(single[1], single[3])
        both = torch.zeros_like(single)
        eng.musetalk_infer([(aid, index[s], Bf, feats[s].data_ptr(), both[s].data_ptr()) for s in range(S)])
        again = torch.zeros_like(single)
        eng.musetalk_infer([(aid, index[s], Bf, feats[s].data_ptr(), again[s].data_ptr()) for s in range(S)])
        assert torch.equal(both, again), "a repeated call must be bit-identical"
        assert torch.equal(both[1], both[3])
        d = (both.to(torch.int16) - single.to(torch.int16)).abs()
        mse = float((d.float() ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        print(f"[mt full size] 64-frame call vs 16-frame calls: max diff {int(d.max())} LSB, differing bytes {float((d != 0).float().mean()):.2e}, PSNR {psnr:.1f} dB")
        assert int(d.max()) <= 2 and psnr >= 50.0
    finally:
        eng.close()


@pytest.mark.gpu
def test_infer_b16_end_to_end_vs_oracle():
    """BASELINE.json configs[2] at its batch size: ONE 16-frame ltk_musetalk_infer call (latent gather by mirror_index across the
    ping-pong turn, PE, U-Net, VAE decode, uint8 BGR) against the oracle run end to end on the same latents / features -
    frames out vs frames out, asserted: PSNR >= 40 dB, max-abs <= 6 LSB, >= 99 % of the bytes within +-2 LSB."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    from oracle import paste_oracle
    Bf, n = 16, 5
    unet_sd, vae_sd = synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict()
    eng = Engine(0)
    try:
        eng.load_musetalk(unet_sd, vae_sd, max_frames=Bf)
        lats = synth.musetalk_latents(n)
        frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(360, 640), box=160, seed=2)
        aid = eng.register_musetalk_avatar(lats, frames, [(240, 100, 400, 280)] * n, [np.full((255, 220, 3), 255, np.uint8)] * n,
                                           [(210, 60, 430, 315)] * n)
        feat = synth.musetalk_whisper_feats(Bf, seed=77)
        d_feat = torch.from_numpy(feat).cuda()
        d_pred = torch.zeros(Bf, 256, 256, 3, dtype=torch.uint8, device="cuda")
        index = 2                                                  # bank positions 2,3,4,4,3,2,1,0,0,1,... (n = 5)
        eng.musetalk_infer([(aid, index, Bf, d_feat.data_ptr(), d_pred.data_ptr())])
        got = d_pred.cpu().numpy()
        usd = {k: torch.from_numpy(v) for k, v in unet_sd.items()}
        vsd = {k: torch.from_numpy(v) for k, v in vae_sd.items()}
        lat = np.concatenate([lats[paste_oracle.mirror_index(n, index + i)] for i in range(Bf)])
        with torch.no_grad():
            ref_lat = M.unet_forward(usd, torch.from_numpy(lat), M.positional_encoding(torch.from_numpy(feat)))
            ref_img = M.vae_decode(vsd, ref_lat / M.VAE_SCALING)
        ref = ((ref_img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).round().astype(np.uint8)[..., ::-1]
        worst = 99.0
        for i in range(Bf):
            di = got[i].astype(np.float64) - ref[i].astype(np.float64)
            mse = float((di * di).mean())
            worst = min(worst, 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse))
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        print(f"[mt B=16] worst frame PSNR {worst:.2f} dB, max {d.max()} LSB, within+-2: {float((d <= 2).mean()):.5f}")
        assert worst >= 40.0 and d.max() <= 6 and float((d <= 2).mean()) >= 0.99
    finally:
        eng.close()


@pytest.mark.gpu
def test_paste_blend_vs_reference_golden(golden_dir):
    """ltk_paste_blend against what the reference's own MuseReal.paste_back_frame + get_image_blending produced
    (tests/golden/musetalk_plugin_golden.npz, written by oracle/gen_golden_musetalk.py::pin_plugin from
    avatars/musetalk_avatar.py:154-164 + myutil.py:4-25): CRC-exact on the four composite cases - soft mask on a growing
    face box, a crop box touching the frame's left and bottom edges, a shrinking face box, a hard 0/255 mask - through the
    per-frame entry and through the batched egress entry the plugin's paste_back_frame uses."""
    import os
    import zlib
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from livetalking_amd.engine import Engine
    from livetalking_amd.egress import SRC_MUSETALK, DeviceEgress, FrameGroup
    from oracle import paste_oracle
    g = np.load(os.path.join(golden_dir, "musetalk_plugin_golden.npz"))
    frames, masks, face_boxes, crop_boxes, preds = synth.musetalk_blend_avatar()
    eng = Engine(0)
    try:
        aid = eng.register_musetalk_avatar(synth.musetalk_latents(4), frames, face_boxes, masks, crop_boxes)
        d_pred = torch.from_numpy(preds).cuda()
        H, W = frames[0].shape[:2]
        for i in range(4):
            out = np.empty((H, W, 3), np.uint8)
            eng.paste_blend(aid, i, d_pred[i].data_ptr(), out)
            ref = paste_oracle.paste_blend_frame(preds[i], frames[i], face_boxes[i], masks[i], crop_boxes[i])
            dd = np.abs(out.astype(int) - ref.astype(int))
            assert np.array_equal(out, ref), (i, int(dd.max()), int((dd != 0).sum()))
            assert zlib.crc32(out.tobytes()) == int(g["blend_crc"][i]), i
        eg = DeviceEgress(eng, H, W, SRC_MUSETALK, aid, fmt="bgr24", watermark=None)
        try:
            items = list(d_pred.unbind(0))
            FrameGroup.attach(items, d_pred, idx=[0, 1, 2, 3])
            for i in range(4):                      # the first call composites all four (ltk_egress_batch), one pinned copy
                o = eg.speaking_frame_of(items[i], i)
                assert zlib.crc32(np.ascontiguousarray(o).tobytes()) == int(g["blend_crc"][i]), ("batched", i)
        finally:
            eg.close()
    finally:
        eng.close()


@pytest.mark.gpu
def test_musetalk_graph_replay_equals_eager_launches(mt):
    """Knob GRAPH (default on): the 436-launch U-Net + VAE program of a given frame count is captured as ONE hipGraph the
    second time it is seen and replayed from then on; the kernels with per-call pointers (latent / token gather, frame
    writer) stay outside it.  Four calls with different bank positions, audio features and output tensors - eager, capture,
    two replays - must give byte for byte what the same calls give launch by launch (GRAPH=0)."""
    from livetalking_amd.engine import Engine
    eng, usd, vsd = mt
    n = 4
    lats = synth.musetalk_latents(n)
    frames, masks, face_boxes, crop_boxes, _ = synth.musetalk_blend_avatar()
    aid = eng.register_musetalk_avatar(lats, frames, face_boxes, masks, crop_boxes)
    feats = [torch.from_numpy(synth.musetalk_whisper_feats(B, seed=40 + k)).cuda() for k in range(4)]

    def run_all():
        outs = []
        for k in range(4):
            pred = torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device="cuda")
            eng.musetalk_infer([(aid, 1 + 2 * k, B, feats[k].data_ptr(), pred.data_ptr())])
            outs.append(pred)
        return outs

    try:
        Engine.set_knob("GRAPH", 0)
        before = eng.program_graph_count()          # (graphs of earlier tests on this engine stay until a run with the knob on drops them)
        eager = run_all()
        assert eng.program_graph_count() == before, "a pass was captured with knob GRAPH off"
        Engine.set_knob("GRAPH", 1)
        replay = run_all()                          # eager (the knob change dropped every graph), capture, two replays
        assert eng.program_graph_count() == 1, "the MuseTalk pass was not captured"
        for k in range(4):
            assert torch.equal(eager[k], replay[k]), f"call {k}: graph replay differs from the eager launches"
        assert not torch.equal(eager[0], eager[1])
    finally:
        Engine.set_knob("GRAPH", 1)


@pytest.mark.gpu
def test_lds_staged_self_attention_equals_per_wave_attention(mt):
    """Knob ATTN_LDS (round 6, nn_kernels.hip attn_lds_kernel): the self-attentions of the 32^2 / 16^2 levels share their key /
    value tiles between a block's four query tiles through LDS.  Same arithmetic in the same order as attn_kernel: the frames of
    a call must be byte for byte the frames of the knob off."""
    from livetalking_amd.engine import Engine
    eng, usd, vsd = mt
    n = 4
    lats = synth.musetalk_latents(n)
    frames, masks, face_boxes, crop_boxes, _ = synth.musetalk_blend_avatar()
    aid = eng.register_musetalk_avatar(lats, frames, face_boxes, masks, crop_boxes)
    feats = torch.from_numpy(synth.musetalk_whisper_feats(B, seed=77)).cuda()

    def run():
        pred = torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            eng.musetalk_infer([(aid, 1, B, feats.data_ptr(), pred.data_ptr())])
        return pred

    try:
        Engine.set_knob("ATTN_LDS", 0)
        off = run()
        Engine.set_knob("ATTN_LDS", 1)
        on = run()
        assert torch.equal(on, off), f"differing bytes: {int((on != off).sum())}"
        assert int(on.max()) > 0
    finally:
        Engine.set_knob("ATTN_LDS", 1)


@pytest.mark.gpu
def test_one_pass_groupnorm_of_the_large_maps(mt):
    """Knob GN_COOP (round 6, nn_kernels.hip gn_coop_kernel): the GroupNorms of the VAE's 64^2 .. 256^2 maps read their tensor once - a block
    keeps its slice in registers, the blocks of an (image, channel block) exchange partial sums through global memory (slots reset at the head
    of every pass, also inside the replayed graph).  Another fp32 summation order than gn_stats + gn_apply: frames within 1 LSB of the knob off;
    deterministic: eight calls (eager, capture, replays) give the same bytes."""
    from livetalking_amd.engine import Engine
    eng, usd, vsd = mt
    n = 4
    lats = synth.musetalk_latents(n)
    frames, masks, face_boxes, crop_boxes, _ = synth.musetalk_blend_avatar()
    aid = eng.register_musetalk_avatar(lats, frames, face_boxes, masks, crop_boxes)
    feats = torch.from_numpy(synth.musetalk_whisper_feats(B, seed=78)).cuda()

    def run(times):
        outs = []
        for _ in range(times):
            pred = torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device="cuda")
            eng.musetalk_infer([(aid, 1, B, feats.data_ptr(), pred.data_ptr())])
            outs.append(pred)
        return outs

    try:
        Engine.set_knob("GN_COOP", 0)
        off = run(2)[-1]
        Engine.set_knob("GN_COOP", 1)
        on = run(8)
        for k in range(1, 8):
            assert torch.equal(on[k], on[0]), f"call {k} differs from call 0: {int((on[k] != on[0]).sum())} bytes"
        d = (on[0].int() - off.int()).abs()
        print(f"[mt] one-pass GroupNorm vs two-pass: max diff {int(d.max())} LSB, differing bytes {float((d != 0).float().mean()):.2e}")
        assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 0.1
        assert int(on[0].max()) > 0
    finally:
        Engine.set_knob("GN_COOP", 1)
