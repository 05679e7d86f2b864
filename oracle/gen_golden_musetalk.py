#!/usr/bin/env python3
"""Pin the MuseTalk HOST-side restatements against the reference's own code (run in the build container).

TEST INFRASTRUCTURE ONLY.   python -m oracle.gen_golden_musetalk [--ref /root/reference]

What runs from the reference, unmodified (imported, never copied):
  avatars.musetalk.models.unet.PositionalEncoding            (unet.py:12-27; `diffusers` stubbed: only the class import)
  avatars.audio_features.whisper.WhisperASR._feature2chunks  (whisper.py:35-56) + BaseASR._get_sliced_feature
                                                             (base_asr.py:91-133), via WhisperASR.run_step's arguments
  avatars.musetalk_avatar: mirror_index use                  (utils/image.py:26-32)
  avatars.musetalk.models.syncnet.ResnetBlock2D              (syncnet.py:71-139, the in-tree twin of diffusers' ResnetBlock2D
                                                             without temb, incl. the asymmetric-pad stride-2 downsample)
  avatars.musetalk.whisper.whisper.model.MultiHeadAttention  (model.py:57-100: the same multi-head attention arithmetic as
                                                             diffusers' Attention; pins head split + scaling)
  avatars.musetalk.myutil.get_image_blending                 (myutil.py:4-25) and
  avatars.musetalk_avatar.MuseReal.paste_back_frame          (musetalk_avatar.py:154-164), under a cv2 stub that carries the
                                                             restated leaves resize / cvtColor / blendLinear (paste_oracle.py):
                                                             the same recipe gen_golden.py uses for LipReal.paste_back_frame
  avatars.musetalk_avatar.MuseReal.inference_batch           (musetalk_avatar.py:130-152) and
  avatars.musetalk.models.vae.VAE.decode_latents             (vae.py:96-108), with the reference's own PositionalEncoding and
                                                             `unet.model` / `vae.vae.decode` replaced by the oracle's network
                                                             functions: pins the CONTROL FLOW (latent gather by mirror_index
                                                             across the ping-pong turn, timesteps=[0], PE, 1/scaling_factor,
                                                             x/2+0.5 clamp, round, RGB->BGR flip), not the networks
The U-Net / VAE graph itself lives in `diffusers` (absent): see oracle/musetalk_oracle.py (PARITY UNPINNED as a whole).
Writes tests/golden/musetalk_host_golden.npz, musetalk_blocks_golden.npz and musetalk_plugin_golden.npz after asserting that the oracle restatements agree.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

import synth_inputs as synth  # noqa: E402
from oracle import musetalk_oracle, paste_oracle, whisper_oracle  # noqa: E402
from oracle import ref_loop as gen_golden  # noqa: E402  (install_stubs / _stub)


def pin_plugin(out_dir: str):
    """MuseReal.paste_back_frame + get_image_blending and MuseReal.inference_batch + VAE.decode_latents, run from the
    reference, against paste_oracle.paste_blend_frame and musetalk_oracle.inference_batch."""
    cv2 = sys.modules["cv2"]
    cv2.COLOR_BGR2GRAY = 6
    cv2.COLOR_BGR2RGB = 4
    calls = {"cvtColor": 0, "blendLinear": 0, "resize": 0}

    def cvt(img, code):
        assert code == cv2.COLOR_BGR2GRAY
        calls["cvtColor"] += 1
        return paste_oracle.cvt_bgr2gray_u8(img)

    def blend(a, b, w1, w2):
        assert a.shape == b.shape and a.dtype == np.uint8 and w1.dtype == np.float32 and w1.shape == a.shape[:2] == w2.shape
        calls["blendLinear"] += 1
        return paste_oracle.blend_linear_u8(a, b, w1, w2)

    inner_resize = cv2.resize

    def resize(src, dsize, *a, **k):
        assert not a and not k                                   # default INTER_LINEAR (musetalk_avatar.py:159)
        calls["resize"] += 1
        return inner_resize(src, dsize)

    cv2.cvtColor, cv2.blendLinear, cv2.resize = cvt, blend, resize
    cv2.VideoCapture = object
    cv2.INTER_LANCZOS4 = 4
    import avatars.musetalk_avatar as ref_plugin
    from avatars.musetalk.myutil import get_image_blending
    from avatars.musetalk.models.vae import VAE
    from avatars.musetalk.models.unet import PositionalEncoding
    from utils.image import mirror_index as ref_mirror

    # ---- (a) the composite
    frames, masks, face_boxes, crop_boxes, preds = synth.musetalk_blend_avatar()
    mr = ref_plugin.MuseReal.__new__(ref_plugin.MuseReal)
    mr.frame_list_cycle, mr.mask_list_cycle = frames, masks
    mr.coord_list_cycle, mr.mask_coords_list_cycle = face_boxes, crop_boxes
    crcs, subs = [], []
    for i in range(4):
        before = frames[i].copy()
        # what inference_batch hands over: a uint8 array; and a float array (a foreign producer): astype truncates
        ref_frame = mr.paste_back_frame(preds[i], i)
        ref_float = mr.paste_back_frame(preds[i].astype(np.float32) + 0.75, i)
        assert np.array_equal(frames[i], before), "the cached frame must survive (ori_frame is a copy)"
        mine = paste_oracle.paste_blend_frame(preds[i], frames[i], face_boxes[i], masks[i], crop_boxes[i])
        assert ref_frame.dtype == np.uint8 and ref_frame.shape == frames[i].shape and ref_frame.flags["C_CONTIGUOUS"]
        assert np.array_equal(ref_frame, mine), f"paste_blend_frame restatement drifted (frame {i})"
        assert np.array_equal(ref_float, mine), f"astype(uint8) truncation (frame {i})"
        # get_image_blending called directly, on a mask whose channels DIFFER: the cvtColor leg is really taken
        rng = np.random.default_rng(100 + i)
        cmask = rng.integers(0, 256, masks[i].shape, dtype=np.uint8)
        x1, y1, x2, y2 = face_boxes[i]
        face = paste_oracle.resize_linear_u8(preds[i], (x2 - x1, y2 - y1))
        ref_c = get_image_blending(frames[i].copy(), face, face_boxes[i], cmask, crop_boxes[i])
        gray3 = np.repeat(paste_oracle.cvt_bgr2gray_u8(cmask)[:, :, None], 3, axis=2)
        assert np.array_equal(ref_c, paste_oracle.paste_blend_frame(preds[i], frames[i], face_boxes[i], gray3, crop_boxes[i]))
        xs, ys, xe, ye = crop_boxes[i]
        crcs.append(zlib.crc32(ref_frame.tobytes()))
        subs.append(ref_frame[ys:ye:4, xs:xe:4][:60, :55].copy())
    assert calls["cvtColor"] == 12 and calls["blendLinear"] == 12 and calls["resize"] == 8, calls
    print("composite: reference MuseReal.paste_back_frame / get_image_blending == oracle on 4 frames "
          "(soft mask on a growing box, crop box on the frame edge, shrinking box, hard mask)")

    # ---- (b) inference_batch control flow: the reference's method and VAE.decode_latents around the oracle's networks
    unet_sd = {k: torch.from_numpy(v) for k, v in synth.musetalk_unet_state_dict().items()}
    vae_sd = {k: torch.from_numpy(v) for k, v in synth.vae_decoder_state_dict().items()}
    n, B, index = 3, 4, 1                      # bank frames 1, 2, 2, 1: across the ping-pong turn
    lats = [torch.from_numpy(x) for x in synth.musetalk_latents(n)]
    feats = synth.musetalk_whisper_feats(B, seed=31)
    seen = {}

    class _UNetModel:
        dtype = torch.float32

        def __call__(self, latent, timesteps, encoder_hidden_states=None):
            seen["timesteps"] = timesteps.clone()
            seen["latent"] = latent.clone()
            return types.SimpleNamespace(sample=musetalk_oracle.unet_forward(unet_sd, latent, encoder_hidden_states,
                                                                            timestep=int(timesteps[0])))

    class _VaeModel:
        dtype = torch.float32
        config = types.SimpleNamespace(scaling_factor=musetalk_oracle.VAE_SCALING)

        def decode(self, z):
            seen["z"] = z.clone()
            return types.SimpleNamespace(sample=musetalk_oracle.vae_decode(vae_sd, z))

    vae = VAE.__new__(VAE)                     # the ctor calls AutoencoderKL.from_pretrained; decode_latents needs these two
    vae.vae, vae.scaling_factor = _VaeModel(), _VaeModel.config.scaling_factor
    mr = ref_plugin.MuseReal.__new__(ref_plugin.MuseReal)
    mr.batch_size = B
    mr.input_latent_list_cycle = lats
    mr.unet = types.SimpleNamespace(device=torch.device("cpu"), model=_UNetModel())
    mr.vae, mr.pe, mr.timesteps = vae, PositionalEncoding(d_model=384), torch.tensor([0])
    with torch.no_grad():
        ref_pred = mr.inference_batch(index, [feats[i] for i in range(B)])
        mine = musetalk_oracle.inference_batch(unet_sd, vae_sd, lats, index, B, feats)
    assert ref_pred.shape == (B, 256, 256, 3) and ref_pred.dtype == np.uint8
    assert int(seen["timesteps"][0]) == 0 and len(seen["timesteps"]) == 1
    order = [ref_mirror(n, index + i) for i in range(B)]
    assert order == [1, 2, 2, 1]
    assert all(torch.equal(seen["latent"][i], lats[order[i]][0]) for i in range(B))
    d = np.abs(ref_pred.astype(np.int32) - mine.astype(np.int32))
    assert d.max() == 0, f"inference_batch restatement drifted: max {d.max()} LSB on {int((d != 0).sum())} bytes"
    print(f"inference_batch: reference MuseReal.inference_batch + VAE.decode_latents == oracle, B={B} @ index {index} of {n} "
          f"(bank order {order}), bit-identical")
    np.savez_compressed(
        os.path.join(out_dir, "musetalk_plugin_golden.npz"),
        blend_generator="synth_inputs.musetalk_blend_avatar()", blend_crc=np.asarray(crcs, dtype=np.uint32),
        blend_sub=np.stack(subs), infer_n=n, infer_batch=B, infer_index=index, infer_feat_seed=31,
        infer_order=np.asarray(order), infer_pred_sub=np.ascontiguousarray(ref_pred[:, ::4, ::4]),
        infer_pred_mean=ref_pred.reshape(B, -1).mean(axis=1))
    print("wrote musetalk_plugin_golden.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    gen_golden.install_stubs()
    gen_golden._stub("diffusers", UNet2DConditionModel=object, AutoencoderKL=object)
    gen_golden._stub("torchvision")
    gen_golden._stub("torchvision.transforms", Normalize=lambda **k: None)
    sys.path.insert(0, args.ref)
    args.out = os.path.abspath(args.out)
    os.makedirs(args.out, exist_ok=True)
    os.chdir(tempfile.mkdtemp(prefix="ltk_golden_mt_"))

    # ---- PositionalEncoding (unet.py:12-27)
    from avatars.musetalk.models.unet import PositionalEncoding
    pe = PositionalEncoding(d_model=384)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 50, 384)).astype(np.float32))
    ref_pe = pe(x).numpy()
    mine = musetalk_oracle.positional_encoding(x).numpy()
    assert np.abs(ref_pe - mine).max() < 1e-6, "PositionalEncoding restatement drifted"
    pe_table = pe(torch.zeros(1, 50, 384)).numpy()[0]

    # ---- whisper chunk slicing (whisper.py:35-56,71-73; base_asr.py:91-133)
    from avatars.audio_features.whisper import WhisperASR
    asr = WhisperASR.__new__(WhisperASR)          # no queues / opt needed for the slicing helper
    feat = np.arange(1500 * 5 * 384, dtype=np.float32).reshape(1500, 5, 384)
    chunks = asr._feature2chunks(feature_array=feat, batch_size=16, audio_feat_win=[0, 5], start=10 / 2, feature_idx_multiplier=2)
    ref_chunks = np.stack(chunks)
    mine_chunks = np.stack(whisper_oracle.feature2chunks(feat, 16, 10))
    assert ref_chunks.shape == (16, 50, 384) and np.array_equal(ref_chunks, mine_chunks), "chunk slicing restatement drifted"
    rows = (ref_chunks[:, :, 0] // (5 * 384)).astype(np.int32)[:, ::5]      # encoder row of every 5-state group

    # ---- edge: clamping at the end of the feature array (short arrays)
    short = feat[:40]
    ref_short = np.stack(asr._feature2chunks(feature_array=short, batch_size=16, audio_feat_win=[0, 5], start=5.0, feature_idx_multiplier=2))
    assert np.array_equal(ref_short, np.stack(whisper_oracle.feature2chunks(short, 16, 10)))
    rows_short = (ref_short[:, :, 0] // (5 * 384)).astype(np.int32)[:, ::5]

    np.savez_compressed(os.path.join(args.out, "musetalk_host_golden.npz"), pe_table=pe_table.astype(np.float32), chunk_rows=rows,
                        chunk_rows_short=rows_short)
    print("wrote musetalk_host_golden.npz: pe_table", pe_table.shape, "chunk_rows", rows.shape, rows[0], rows[-1])

    pin_plugin(args.out)

    # ---- ResnetBlock2D + asymmetric-pad downsample: the reference's in-tree implementation (syncnet.py:71-139).  The module
    # imports two diffusers classes for its attention block; they are stubbed, ResnetBlock2D itself is plain torch.
    gen_golden._stub("diffusers.models")
    gen_golden._stub("diffusers.models.attention", Attention=object, FeedForward=object)
    gen_golden._stub("diffusers.utils")
    gen_golden._stub("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    from avatars.musetalk.models.syncnet import ResnetBlock2D
    torch.manual_seed(7)
    blocks = {"a": ResnetBlock2D(32, 64, norm_num_groups=32, eps=1e-6, downsample_factor=1),       # with conv_shortcut
              "b": ResnetBlock2D(64, 64, norm_num_groups=32, eps=1e-6, downsample_factor=2)}       # identity skip + downsample
    out = {}
    with torch.no_grad():
        for name, blk in blocks.items():
            for prm in blk.parameters():               # default init leaves the norms at (1, 0): randomise everything
                prm.copy_(torch.randn_like(prm) * (0.3 if prm.dim() == 1 else (2.0 / prm[0].numel()) ** 0.5))
            for nm, prm in blk.named_parameters():
                if nm.startswith("norm") and nm.endswith("weight"):
                    prm.add_(1.0)
            cin = 32 if name == "a" else 64
            xin = torch.randn(2, cin, 13, 10)          # odd height: the bottom pad row matters
            y = blk.eval()(xin)
            sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
            # oracle: resnet() + downsample_asym() under the same tensor names
            osd = {f"r.{k}": v for k, v in sd.items() if not k.startswith("downsample_conv")}
            mine = musetalk_oracle.resnet(osd, "r", xin, None, 32, 1e-6)
            if name == "b":
                osd["d.weight"], osd["d.bias"] = sd["downsample_conv.weight"], sd["downsample_conv.bias"]
                mine = musetalk_oracle.downsample_asym(osd, "d", mine)
            err = float((mine - y).abs().max())
            assert mine.shape == y.shape and err < 2e-5, f"ResnetBlock2D restatement drifted ({name}: {err})"
            out[f"{name}_x"] = xin.numpy()
            out[f"{name}_y"] = y.numpy()
            for k, v in sd.items():
                out[f"{name}_sd.{k}"] = v.numpy()
            print(f"ResnetBlock2D {name}: out {tuple(y.shape)} max|restatement - reference| = {err:.2e}")
    # ---- multi-head attention: the in-tree OpenAI Whisper MultiHeadAttention (avatars/musetalk/whisper/whisper/model.py:57-100)
    # is the same operation as diffusers' Attention (separate q/k/v/out Linears, heads = contiguous channel slices,
    # softmax(q k^T d^-0.5) v); it pins the head split and scaling of musetalk_oracle.attention for self- and cross-attention.
    gen_golden._stub("ffmpeg")                      # whisper/audio.py imports it for file decoding only
    from avatars.musetalk.whisper.whisper.model import MultiHeadAttention
    torch.manual_seed(11)
    mha = MultiHeadAttention(128, 8).eval()
    with torch.no_grad():
        for prm in mha.parameters():
            prm.copy_(torch.randn_like(prm) * (0.1 if prm.dim() == 1 else (1.0 / prm.shape[1]) ** 0.5))
        xq = torch.randn(2, 37, 128)
        xc = torch.randn(2, 50, 128)
        y_self = mha(xq)
        y_cross = mha(xq, xa=xc)
        asd = {"m.to_q.weight": mha.query.weight, "m.to_q.bias": mha.query.bias, "m.to_k.weight": mha.key.weight,
               "m.to_v.weight": mha.value.weight, "m.to_v.bias": mha.value.bias, "m.to_out.0.weight": mha.out.weight,
               "m.to_out.0.bias": mha.out.bias}
        e1 = float((musetalk_oracle.attention(asd, "m", xq, xq, 8) - y_self).abs().max())
        e2 = float((musetalk_oracle.attention(asd, "m", xq, xc, 8) - y_cross).abs().max())
        assert e1 < 2e-5 and e2 < 2e-5, f"attention restatement drifted ({e1}, {e2})"
        print(f"MultiHeadAttention: self {e1:.2e}, cross {e2:.2e}")
        out.update({"mha_xq": xq.numpy(), "mha_xc": xc.numpy(), "mha_y_self": y_self.numpy(), "mha_y_cross": y_cross.numpy()})
        for k, v in asd.items():
            out["mha_sd." + k] = v.detach().numpy()
    # ---- Transformer2D composition: the reference's in-tree AttentionBlock2D (syncnet.py:142-181) = GroupNorm(eps 1e-6) -> 1x1
    # conv_in -> LayerNorm -> self-attention + x -> LayerNorm -> GEGLU feed-forward + x -> 1x1 conv_out -> + input, i.e. diffusers'
    # Transformer2DModel / BasicTransformerBlock without the cross-attention sub-block.  Its two diffusers leaf classes are
    # replaced by (a) the reference's own MultiHeadAttention pinned above and (b) a literal GEGLU feed-forward; what is pinned
    # is the COMPOSITION that musetalk_oracle.transformer2d(cross=False) restates.
    import sys as _sys
    att_mod = _sys.modules["diffusers.models.attention"]

    class _Attn(torch.nn.Module):
        def __init__(self, query_dim, heads=8, dim_head=None, dropout=0.0, bias=True):
            super().__init__()
            self.mha = MultiHeadAttention(query_dim, heads)

        def forward(self, x, attention_mask=None):
            out = self.mha(x)
            return out[0] if isinstance(out, tuple) else out

    class _FF(torch.nn.Module):
        def __init__(self, dim, dropout=0.0, activation_fn="geglu"):
            super().__init__()
            assert activation_fn == "geglu"
            self.proj = torch.nn.Linear(dim, dim * 8)
            self.out = torch.nn.Linear(dim * 4, dim)

        def forward(self, x):
            a, gate = self.proj(x).chunk(2, dim=-1)
            return self.out(a * torch.nn.functional.gelu(gate))

    att_mod.Attention, att_mod.FeedForward = _Attn, _FF
    _sys.modules["diffusers.utils.import_utils"].is_xformers_available = lambda: True
    import importlib
    import avatars.musetalk.models.syncnet as syncnet
    importlib.reload(syncnet)
    torch.manual_seed(13)
    blk = syncnet.AttentionBlock2D(128).eval()
    with torch.no_grad():
        for nm, prm in blk.named_parameters():
            prm.copy_(torch.randn_like(prm) * (0.2 if prm.dim() == 1 else (1.0 / prm[0].numel()) ** 0.5))
            if "norm" in nm and nm.endswith("weight"):
                prm.add_(1.0)
        xa = torch.randn(2, 128, 6, 5)
        ya = blk(xa)
        m = blk.attn.mha
        tsd = {"t.norm.weight": blk.norm1.weight, "t.norm.bias": blk.norm1.bias,
               "t.proj_in.weight": blk.conv_in.weight, "t.proj_in.bias": blk.conv_in.bias,
               "t.proj_out.weight": blk.conv_out.weight, "t.proj_out.bias": blk.conv_out.bias,
               "t.transformer_blocks.0.norm1.weight": blk.norm2.weight, "t.transformer_blocks.0.norm1.bias": blk.norm2.bias,
               "t.transformer_blocks.0.norm3.weight": blk.norm3.weight, "t.transformer_blocks.0.norm3.bias": blk.norm3.bias,
               "t.transformer_blocks.0.attn1.to_q.weight": m.query.weight, "t.transformer_blocks.0.attn1.to_q.bias": m.query.bias,
               "t.transformer_blocks.0.attn1.to_k.weight": m.key.weight,
               "t.transformer_blocks.0.attn1.to_v.weight": m.value.weight, "t.transformer_blocks.0.attn1.to_v.bias": m.value.bias,
               "t.transformer_blocks.0.attn1.to_out.0.weight": m.out.weight, "t.transformer_blocks.0.attn1.to_out.0.bias": m.out.bias,
               "t.transformer_blocks.0.ff.net.0.proj.weight": blk.ff.proj.weight, "t.transformer_blocks.0.ff.net.0.proj.bias": blk.ff.proj.bias,
               "t.transformer_blocks.0.ff.net.2.weight": blk.ff.out.weight, "t.transformer_blocks.0.ff.net.2.bias": blk.ff.out.bias}
        mine = musetalk_oracle.transformer2d(tsd, "t", xa, None, cross=False, heads=8, groups=32, gn_eps=1e-6)
        e3 = float((mine - ya).abs().max())
        assert mine.shape == ya.shape and e3 < 5e-5, f"Transformer2D composition drifted ({e3})"
        print(f"AttentionBlock2D (Transformer2D composition): max|restatement - reference| = {e3:.2e}")
        out.update({"t2d_x": xa.numpy(), "t2d_y": ya.numpy()})
        for k, v in tsd.items():
            out["t2d_sd." + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(args.out, "musetalk_blocks_golden.npz"), **out)
    print("wrote musetalk_blocks_golden.npz")


if __name__ == "__main__":
    main()
