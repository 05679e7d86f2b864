"""End-to-end through the plugin surface (module functions + LipReal) on the GPU:
PCM chunks in -> composited uint8 frames out, compared with the oracle chain
(mel_oracle -> plugin_oracle -> paste_oracle), following the call order of the
reference's render/inference/process_frames loops (avatars/base_avatar.py:337-376,
:433, :487-494)."""
import argparse

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import mel_oracle, paste_oracle, plugin_oracle, synth  # noqa: E402


@pytest.mark.gpu
def test_lipreal_headless_render_loop():
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.hostshim import mirror_index

    sd_np = synth.wav2lip_state_dict(1234)
    model = plugin.load_model(None, state_dict=sd_np, max_frames=8)
    B = 4
    plugin.warm_up(B, model, 256)
    avatar = synth.wav2lip_avatar(n_frames=5, full_hw=(360, 640), box=160, seed=0)
    frames, faces, coords = avatar
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    sess = plugin.LipReal(opt, model, avatar)
    assert sess.get_avatar_length() == 5 and hasattr(sess, "asr")

    audio = synth.synthetic_audio(2.0)
    n_steps = 3
    # warm_up consumed l+r chunks of silence (queue was empty); now feed speech
    for c in range(n_steps * 2 * B):
        sess.put_audio_frame(audio[c * 320:(c + 1) * 320], {})
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    pcm_hist = [np.zeros(320, np.float32)] * 20
    index = 0
    for step in range(n_steps):
        sess.asr.run_step()                                   # render thread
        feat = sess.asr.feat_queue.get(timeout=1)             # inference thread
        audio_frames = [sess.asr.output_queue.get() for _ in range(2 * B)]
        assert len(audio_frames) == 2 * B      # audio out lags features by the r look-ahead chunks
        pred = sess.inference_batch(index, feat)
        assert len(pred) == B
        # oracle for the same step
        pcm_hist = pcm_hist + [audio[c * 320:(c + 1) * 320] for c in range(step * 2 * B, (step + 1) * 2 * B)]
        wav = np.concatenate(pcm_hist)
        ref_feats = mel_oracle.mel_chunks(wav, len(pcm_hist))
        assert float(np.abs(feat.cpu().numpy() - np.stack(ref_feats)).max()) <= 1e-3
        ref_pred = plugin_oracle.inference_batch(sd, faces, index, B, ref_feats)
        for i, res_frame in enumerate(pred):                  # process thread
            idx = mirror_index(len(frames), index + i)
            out = sess.paste_back_frame(res_frame, idx)
            assert out.dtype == np.uint8 and out.shape == (360, 640, 3) and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
            ref = paste_oracle.paste_back_frame(ref_pred[i], frames[idx], coords[idx])
            d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 6 and (d <= 2).mean() >= 0.995, (step, i, int(d.max()))
            # bit-exact composite given the engine's own uint8 crop
            own = paste_oracle.paste_back_frame(res_frame.cpu().numpy().astype(np.float32), frames[idx], coords[idx])
            assert np.array_equal(out, own)
        index += B
        pcm_hist = pcm_hist[-20:]
    model.engine.close()
