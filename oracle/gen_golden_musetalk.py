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
The U-Net / VAE graph itself lives in `diffusers` (absent): see oracle/musetalk_oracle.py (PARITY UNPINNED as a whole).
Writes tests/golden/musetalk_host_golden.npz and musetalk_blocks_golden.npz after asserting that the oracle restatements agree.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import musetalk_oracle, whisper_oracle  # noqa: E402
from oracle import ref_loop as gen_golden  # noqa: E402  (install_stubs / _stub)


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
