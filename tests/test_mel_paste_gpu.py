"""HIP mel-spectrogram and paste-back kernels vs the oracle and the golden
fixtures made by the reference's own MelASR / paste_back_frame."""
import os
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import mel_oracle, paste_oracle  # noqa: E402
import synth_inputs as synth

MEL_TOL = 1e-3   # normalised mel units (range +-4); SURVEY.md §8c


@pytest.mark.gpu
def test_mel_vs_reference_golden(engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "mel_golden.npz"))
    audio = synth.synthetic_audio(float(g["audio_seconds"]), seed=int(g["audio_seed"]))
    starts = [int(s) for s in g["window_starts"]]
    out = torch.zeros(16, 80, 16, dtype=torch.float32, device="cuda")
    worst = 0.0
    for step in range(3):
        wav = audio[step * 10240: step * 10240 + 16640]
        engine.mel_step(wav, starts, out.data_ptr())
        got = out.cpu().numpy()
        ref = g["ref_chunks"][step]
        worst = max(worst, float(np.abs(got - ref).max()))
    print(f"[mel] max abs err vs reference MelASR.run_step: {worst:.3e}")
    assert worst <= MEL_TOL
    # batch_size=1 configuration (22 chunks, one window at column 16)
    out1 = torch.zeros(1, 80, 16, dtype=torch.float32, device="cuda")
    engine.mel_step(audio[:7040], [16], out1.data_ptr())
    assert float(np.abs(out1.cpu().numpy() - g["ref_chunks_b1"]).max()) <= MEL_TOL


@pytest.mark.gpu
def test_mel_edge_cases(engine):
    rng = np.random.default_rng(3)
    out = torch.zeros(16, 80, 16, dtype=torch.float32, device="cuda")
    starts = mel_oracle.window_starts(52, 10, 10)
    # silence -> clipped floor (-4 everywhere); loud noise; a pure tone
    for name, wav in (("silence", np.zeros(16640, np.float32)),
                      ("noise", rng.standard_normal(16640).astype(np.float32)),
                      ("tone", (0.5 * np.sin(2 * np.pi * 440 * np.arange(16640) / 16000)).astype(np.float32))):
        engine.mel_step(wav, starts, out.data_ptr())
        ref = np.stack(mel_oracle.mel_chunks(wav, 52))
        err = float(np.abs(out.cpu().numpy() - ref).max())
        print(f"[mel] {name}: max abs err {err:.3e}")
        assert err <= MEL_TOL, name


@pytest.mark.gpu
def test_paste_bit_exact(engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "paste_golden.npz"))
    gw = np.load(os.path.join(golden_dir, "wav2lip_golden.npz"))
    hw = tuple(int(v) for v in gw["avatar_hw"])
    frames, faces, coords = synth.wav2lip_avatar(n_frames=int(gw["avatar_frames"]), full_hw=hw,
                                                 box=int(gw["avatar_box"]), seed=int(gw["avatar_seed"]))
    aid = engine.register_avatar(faces, frames, coords)
    B, index = int(gw["batch"]), int(gw["index"])
    # oracle/gen_golden.py paste_pred: deterministic bytes.  (The synthetic crops are transposed VIEWS and np.stack keeps their
    # memory order: without the explicit C-contiguous copy the device would receive a transposed image.)
    pred_u8 = np.ascontiguousarray(np.stack([faces[(i + 1) % len(faces)] for i in range(B)]))
    assert pred_u8.flags["C_CONTIGUOUS"] and pred_u8[0].flags["C_CONTIGUOUS"]
    for i in range(B):
        idx = paste_oracle.mirror_index(len(frames), index + i)
        d_pred = torch.from_numpy(pred_u8[i]).cuda()
        out = np.empty((hw[0], hw[1], 3), dtype=np.uint8)
        engine.paste_back(aid, idx, d_pred.data_ptr(), out)
        assert out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
        assert zlib.crc32(out.tobytes()) == int(g["frame_crc"][i]), f"frame {i} differs from the reference composite"
        ref = paste_oracle.paste_back_frame(pred_u8[i].astype(np.float32), frames[idx], coords[idx])
        assert np.array_equal(out, ref)
    engine.release_avatar(aid)
    # shrinking box, exact-2x box, identity box, box touching the frame border
    frames2, faces2, coords2 = synth.wav2lip_avatar(n_frames=4, full_hw=hw, box=int(g["shrink_box"]), seed=int(g["shrink_seed"]))
    coords2[1] = tuple(int(v) for v in g["shrink_coords1"])
    coords2[2] = (50, 306, 100, 356)          # 256x256: identity
    coords2[3] = (0, 301, hw[1] - 333, hw[1])  # touches top/right border, odd size, upscale
    aid2 = engine.register_avatar(faces2, frames2, coords2)
    for i in range(4):
        d_pred = torch.from_numpy(pred_u8[i % B]).cuda()
        out = np.empty((hw[0], hw[1], 3), dtype=np.uint8)
        engine.paste_back(aid2, i, d_pred.data_ptr(), out)
        ref = paste_oracle.paste_back_frame(pred_u8[i % B].astype(np.float32), frames2[i], coords2[i])
        assert np.array_equal(out, ref), f"shrink case {i}"
        if i < 2:
            assert zlib.crc32(out.tobytes()) == int(g["shrink_crc"][i])
    engine.release_avatar(aid2)
