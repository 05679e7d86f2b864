"""End-to-end through the MuseTalk plugin surface (module functions + MuseReal + WhisperASR) on the GPU: PCM chunks in
-> blended uint8 frames out, following the call order of the reference's render / inference / process_frames loops
(avatars/base_avatar.py:337-376, :433, :487-494), checked against the oracle chain
(whisper_oracle -> musetalk_oracle -> paste_oracle)."""
import argparse

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytest.importorskip("transformers")

import synth_inputs as synth  # noqa: E402
from oracle import musetalk_oracle as M  # noqa: E402
from oracle import paste_oracle, whisper_oracle as WO  # noqa: E402


@pytest.mark.gpu
def test_musereal_headless_render_loop():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import livetalking_amd.avatars.musetalk_avatar as plugin
    from livetalking_amd.hostshim import mirror_index

    unet_sd, vae_sd = synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict()
    wm = WO.tiny_whisper(0)
    B, n = 2, 3
    model = plugin.load_model(unet_sd, vae_sd, wm.encoder.state_dict(), max_frames=B)
    plugin.warm_up(B, model)
    lats = [torch.from_numpy(x) for x in synth.musetalk_latents(n)]
    frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(360, 640), box=160, seed=2)
    coords = [(250, 100, 410, 280)] * n
    crops = [(220, 60, 440, 320)] * n
    mask = np.zeros((260, 220, 3), np.uint8)
    mask[130:, 30:-30] = 200
    masks = [mask] * n
    avatar = (frames, masks, coords, crops, lats)
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    sess = plugin.MuseReal(opt, model, avatar)
    assert sess.get_avatar_length() == n

    audio = synth.synthetic_audio(2.0)
    for c in range(2 * 2 * B):
        sess.put_audio_frame(audio[c * 320:(c + 1) * 320], {})
    usd = {k: torch.from_numpy(v) for k, v in unet_sd.items()}
    vsd = {k: torch.from_numpy(v) for k, v in vae_sd.items()}
    pcm_hist = [np.zeros(320, np.float32)] * 20
    index = 0
    for step in range(2):
        sess.asr.run_step()
        feat = sess.asr.feat_queue.get(timeout=1)
        for _ in range(2 * B):
            sess.asr.output_queue.get()
        pred = sess.inference_batch(index, feat)
        assert len(pred) == B
        pcm_hist = pcm_hist + [audio[c * 320:(c + 1) * 320] for c in range(step * 2 * B, (step + 1) * 2 * B)]
        feat_arr, _, _ = WO.audio2feat(wm, np.concatenate(pcm_hist))
        ref_chunks = np.stack(WO.feature2chunks(feat_arr, B, 10))
        r = float(np.linalg.norm(feat.cpu().numpy() - ref_chunks) / np.linalg.norm(ref_chunks))
        assert r <= 1e-2, r
        with torch.no_grad():
            ref_pred = M.inference_batch(usd, vsd, lats, index, B, ref_chunks)
        for i, res_frame in enumerate(pred):
            idx = mirror_index(n, index + i)
            got_crop = res_frame.cpu().numpy()
            d = np.abs(got_crop.astype(np.int32) - ref_pred[i].astype(np.int32))
            mse = float((d.astype(np.float64) ** 2).mean())
            psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
            assert psnr >= 45.0 and d.max() <= 4, (step, i, psnr, int(d.max()))
            out = sess.paste_back_frame(res_frame, idx)
            assert out.dtype == np.uint8 and out.shape == (360, 640, 3) and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
            own = paste_oracle.paste_blend_frame(got_crop, frames[idx], coords[idx], masks[idx], crops[idx])
            assert np.array_equal(out, own)
        index += B
        pcm_hist = pcm_hist[-20:]
    model.engine.close()
