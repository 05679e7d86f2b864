#!/usr/bin/env python3
"""First-run check of REAL MuseTalk checkpoints on the deployment box (GPU + the reference's own dependencies).

The build container has neither `diffusers` nor any checkpoint, so the engine's U-Net / VAE graphs are validated there
against a restatement (oracle/musetalk_oracle.py; VAE pinned to transformers' VQ-VAE, U-Net wiring unpinned).  This script
closes that gap where the real files exist.  Run it from the LiveTalking checkout, once, before serving:

    python /path/to/repo/scripts/verify_musetalk_checkpoint.py [--models ./models] [--frames 4]

1. key / shape diff: models/musetalkV15/unet.pth and models/sd-vae/diffusion_pytorch_model.{safetensors,bin} against
   tests/golden/musetalk_key_manifest.json (what the engine looks up), after the deprecated-attention-name conversion the
   plugin applies (query/key/value/proj_attn -> to_q/to_k/to_v/to_out.0);
2. numerics, when `diffusers` is importable: the reference's own load path (avatars/musetalk/utils/utils.py:15-31:
   UNet2DConditionModel(**musetalk.json) + AutoencoderKL.from_pretrained) on random latents / whisper features in fp32 on the
   CPU, against ltk_musetalk_forward_host on the GPU: U-Net output relative L2 and decoded-image PSNR, with the tolerances of
   tests/test_musetalk_gpu.py (rel L2 <= 1e-2, PSNR >= 40 dB).
Exit code 0 only if every check that could run passed.
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def load_files(models):
    import torch
    unet = torch.load(os.path.join(models, "musetalkV15", "unet.pth"), map_location="cpu")
    p = os.path.join(models, "sd-vae", "diffusion_pytorch_model.safetensors")
    if os.path.exists(p):
        from safetensors.torch import load_file
        vae = load_file(p)
    else:
        vae = torch.load(p.replace(".safetensors", ".bin"), map_location="cpu")
    return unet, vae


def diff_keys(name, sd, manifest):
    got = {k: list(v.shape) for k, v in sd.items()}
    missing = sorted(set(manifest) - set(got))
    wrong = sorted(k for k in manifest if k in got and got[k] != manifest[k])
    extra = sorted(set(got) - set(manifest))
    print(f"[{name}] {len(manifest)} tensors expected: {len(missing)} missing, {len(wrong)} with another shape, {len(extra)} not used")
    for k in missing[:20]:
        print("   missing:", k, manifest[k])
    for k in wrong[:20]:
        print("   shape:", k, got[k], "expected", manifest[k])
    return not missing and not wrong


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="./models")
    ap.add_argument("--frames", type=int, default=4)
    args = ap.parse_args()
    os.environ.setdefault("LTK_ALLOW_STANDIN", "1")
    from livetalking_amd.avatars.musetalk_avatar import convert_deprecated_vae_attention
    with open(os.path.join(REPO, "tests", "golden", "musetalk_key_manifest.json")) as f:
        man = json.load(f)
    unet_sd, vae_sd = load_files(args.models)
    vae_sd = convert_deprecated_vae_attention(vae_sd)
    ok = diff_keys("unet", unet_sd, man["unet"])
    ok &= diff_keys("vae decoder", {k: v for k, v in vae_sd.items() if k.startswith(("decoder.", "post_quant_conv."))}, man["vae_decoder"])
    ok &= diff_keys("vae encoder", {k: v for k, v in vae_sd.items() if k.startswith(("encoder.", "quant_conv."))}, man["vae_encoder"])
    try:
        import diffusers  # noqa: F401
        import torch
        from diffusers import AutoencoderKL, UNet2DConditionModel
    except Exception as ex:  # noqa: BLE001
        print("diffusers is not importable here (", ex, "): numerics check skipped")
        sys.exit(0 if ok else 1)
    with open(os.path.join(args.models, "musetalkV15", "musetalk.json")) as f:
        cfg = json.load(f)
    unet = UNet2DConditionModel(**cfg)
    unet.load_state_dict(unet_sd)
    unet = unet.float().eval()
    vae = AutoencoderKL.from_pretrained(os.path.join(args.models, "sd-vae")).float().eval()
    B = args.frames
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(B, 8, 32, 32, generator=g) * 0.7
    feat = torch.randn(B, 50, 384, generator=g) * 0.5
    sys.path.insert(0, os.getcwd())
    from avatars.musetalk.models.unet import PositionalEncoding      # the reference's own PE (unet.py:12-27)
    with torch.no_grad():
        ref_u = unet(lat, torch.tensor([0]), encoder_hidden_states=PositionalEncoding(d_model=384)(feat)).sample
        ref_img = vae.decode(ref_u / vae.config.scaling_factor).sample
    from livetalking_amd.engine import Engine
    eng = Engine(0)
    eng.load_musetalk(unet_sd, {k: v for k, v in vae_sd.items() if k.startswith(("decoder.", "post_quant_conv."))}, max_frames=B)
    got_u, got_img, _ = eng.musetalk_forward_host(lat.numpy(), feat.numpy(), want_image=True, want_frames=False)
    rel = float(np.linalg.norm(got_u - ref_u.numpy()) / np.linalg.norm(ref_u.numpy()))
    a = np.clip(got_img / 2 + 0.5, 0, 1) * 255
    b = np.clip(ref_img.numpy() / 2 + 0.5, 0, 1) * 255
    mse = float(((a - b) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    print(f"[numerics] U-Net output rel L2 {rel:.3e} (<= 1e-2), decoded image PSNR {psnr:.1f} dB (>= 40)")
    ok &= rel <= 1e-2 and psnr >= 40.0
    eng.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
