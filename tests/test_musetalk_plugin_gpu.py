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


def _write_musetalk_deployment(root, unet_sd, vae_dec_sd, wm, frames, masks, face_boxes, crop_boxes, lats):
    """The files the reference reads, in its layout: models/musetalkV15/{unet.pth, musetalk.json}, models/sd-vae/
    {config.json, diffusion_pytorch_model.safetensors} (the WHOLE AutoencoderKL under the pre-0.13 attention names, as
    sd-vae-ft-mse ships), models/whisper/ (save_pretrained), data/avatars/<id>/ as avatars/musetalk/genavatar.py:134-156
    writes it (avatars/musetalk/utils/utils.py:15-31, audio2feature.py:15-23, musetalk_avatar.py:69-91)."""
    import json
    import os
    import pickle
    from PIL import Image
    from safetensors.torch import save_file
    from transformers import WhisperFeatureExtractor
    import livetalking_amd.avatars.musetalk_avatar as plugin

    os.makedirs(root / "models" / "musetalkV15")
    torch.save({k: torch.from_numpy(v) for k, v in unet_sd.items()}, root / "models" / "musetalkV15" / "unet.pth")
    cfg = dict(plugin.UNET_TOPOLOGY, _class_name="UNet2DConditionModel", sample_size=64)
    with open(root / "models" / "musetalkV15" / "musetalk.json", "w") as f:
        json.dump(cfg, f)
    os.makedirs(root / "models" / "sd-vae")
    full = {k: torch.from_numpy(v) for k, v in vae_dec_sd.items()}
    full.update({k: torch.from_numpy(v) for k, v in synth.vae_encoder_state_dict().items()})      # encoder.* + quant_conv: must be ignored
    old = {}
    for k, v in full.items():
        for new_name, old_name in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if ".attentions." in k and f".{new_name}." in k:
                k = k.replace(f".{new_name}.", f".{old_name}.")
        old[k] = v.contiguous()
    assert any(".query." in k for k in old)
    save_file(old, str(root / "models" / "sd-vae" / "diffusion_pytorch_model.safetensors"))
    with open(root / "models" / "sd-vae" / "config.json", "w") as f:
        json.dump(dict(plugin.VAE_TOPOLOGY, _class_name="AutoencoderKL", in_channels=3, sample_size=256), f)
    wm.save_pretrained(str(root / "models" / "whisper"))
    WhisperFeatureExtractor().save_pretrained(str(root / "models" / "whisper"))
    adir = root / "data" / "avatars" / "mt1"
    os.makedirs(adir / "full_imgs"); os.makedirs(adir / "mask")
    for i, (fr, m) in enumerate(zip(frames, masks)):
        Image.fromarray(np.ascontiguousarray(fr[..., ::-1])).save(adir / "full_imgs" / f"{i:08d}.png")
        Image.fromarray(np.ascontiguousarray(m[:, :, 0])).save(adir / "mask" / f"{i:08d}.png")     # cv2.imwrite of a 2-D array: a grey PNG
    torch.save([torch.from_numpy(x) for x in lats], adir / "latents.pt")
    with open(adir / "coords.pkl", "wb") as f:
        pickle.dump([list(b) for b in face_boxes], f)
    with open(adir / "mask_coords.pkl", "wb") as f:
        pickle.dump([list(b) for b in crop_boxes], f)
    with open(adir / "avator_info.json", "w") as f:
        json.dump({"avatar_id": "mt1", "version": "v15"}, f)
    return adir


@pytest.mark.gpu
def test_file_format_legs_checkpoints_configs_and_avatar_dir(tmp_path, monkeypatch):
    """M8 end to end on the GPU path: load_model() with NO arguments reads models/musetalkV15/unet.pth (+ musetalk.json,
    checked against the compiled topology), models/sd-vae/diffusion_pytorch_model.safetensors (whole AutoencoderKL,
    deprecated attention names) and ./models/whisper; load_avatar(id) reads a genavatar-layout directory (latents.pt,
    coords.pkl, mask/, mask_coords.pkl, full_imgs/) through the cv2.imread leg AND through the packed .ltkbank leg.
    Whisper step + inference_batch + paste_back_frame must give the same bytes as the state-dict / in-memory path;
    a musetalk.json that asks for another network is refused."""
    import json
    import sys
    import types
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import livetalking_amd.avatars.musetalk_avatar as plugin
    from livetalking_amd import bank
    from livetalking_amd.hostshim import mirror_index

    unet_sd, vae_sd = synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict()
    wm = WO.tiny_whisper(0)
    frames, masks, face_boxes, crop_boxes, _ = synth.musetalk_blend_avatar()
    lats = synth.musetalk_latents(4)
    adir = _write_musetalk_deployment(tmp_path, unet_sd, vae_sd, wm, frames, masks, face_boxes, crop_boxes, lats)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("LTK_DEVICES", "0")
    B = 2
    monkeypatch.setenv("LTK_MT_MAX_FRAMES", str(B))
    # cv2 is not installed in this image: a stand-in whose imread has cv2.imread's semantics for 8-bit PNGs (BGR, 3 channels)
    monkeypatch.setitem(sys.modules, "cv2", types.SimpleNamespace(imread=bank._imread_bgr))

    model = plugin.load_model()                                   # the three checkpoint-file legs + the config checks
    avatar_dir = plugin.load_avatar("mt1")                        # directory leg (glob + integer sort + pickles + torch.load)
    assert len(avatar_dir[0]) == 4 and all(np.array_equal(a, b) for a, b in zip(avatar_dir[0], frames))
    assert all(np.array_equal(a, b) for a, b in zip(avatar_dir[1], masks)) and avatar_dir[1][0].shape[2] == 3
    assert [tuple(c) for c in avatar_dir[2]] == face_boxes and [tuple(c) for c in avatar_dir[3]] == crop_boxes
    bank.pack_avatar_dir(str(adir), kind="musetalk")
    avatar_bank = plugin.load_avatar("mt1")                       # bank.ltkbank now exists: the packed leg
    assert type(avatar_bank[0]).__name__ == "PackedList"

    model2 = plugin.load_model(unet_sd, vae_sd, wm.encoder.state_dict(), max_frames=B)
    audio = synth.synthetic_audio(2.0)
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    outs = []
    for mdl, av in ((model, avatar_dir), (model, avatar_bank), (model2, (frames, masks, face_boxes, crop_boxes, [torch.from_numpy(x) for x in lats]))):
        sess = plugin.MuseReal(opt, mdl, av)
        for c in range(2 * 2 * B):
            sess.put_audio_frame(audio[c * 320:(c + 1) * 320], {})
        got = []
        index = 2                                                 # bank frames 2, 3 then 3, 2: across the turn
        for step in range(2):
            sess.asr.run_step()
            feat = sess.asr.feat_queue.get(timeout=1)
            for _ in range(2 * B):
                sess.asr.output_queue.get()
            pred = sess.inference_batch(index, feat)
            for i, res_frame in enumerate(pred):
                got.append(sess.paste_back_frame(res_frame, mirror_index(4, index + i)).copy())
            index += B
        outs.append(np.stack(got))
    assert outs[2].std() > 5 and not np.array_equal(outs[2][0], frames[2])
    assert np.array_equal(outs[0], outs[2]), "files (directory leg) vs state dicts"
    assert np.array_equal(outs[1], outs[2]), "files (bank leg) vs state dicts"

    # a config that asks for a different network must be refused, naming the field
    with open(tmp_path / "models" / "musetalkV15" / "musetalk.json") as f:
        cfg = json.load(f)
    cfg["block_out_channels"] = [320, 640, 1280]
    with open(tmp_path / "models" / "musetalkV15" / "musetalk.json", "w") as f:
        json.dump(cfg, f)
    with pytest.raises(ValueError, match="block_out_channels"):
        plugin.load_model()
    for m in (model, model2):
        for e in m.engines:
            e.close()
