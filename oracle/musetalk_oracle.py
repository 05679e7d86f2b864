"""Oracle: MuseTalk per-frame generator (conditional U-Net + VAE decoder), plain torch on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What the reference runs for this path (paths relative to the upstream checkout):
  avatars/musetalk_avatar.py:130-152   MuseReal.inference_batch
      latents (B,8,32,32) gathered by mirror_index; PE on the whisper features;
      unet.model(latent, timesteps=[0], encoder_hidden_states=feat).sample; vae.decode_latents
  avatars/musetalk/models/unet.py:12-27   PositionalEncoding (sinusoidal, d_model 384)
  avatars/musetalk/models/unet.py:36-46   diffusers.UNet2DConditionModel(**musetalk.json)
  avatars/musetalk/models/vae.py:96-108   decode_latents: latents/scaling_factor -> AutoencoderKL.decode
      -> (x/2+0.5).clamp(0,1) -> NHWC -> (x*255).round().astype(uint8) -> RGB->BGR

THIRD-PARTY ARITHMETIC, ABSENT HERE: `diffusers` is an unpinned requirement of the reference
(requirements.txt:41), is not vendored under the checkout and is not installed in this
container; the model config `models/musetalkV15/musetalk.json` and every checkpoint are not in
the tree either.  This module therefore RESTATES the published diffusers
algorithm for
  * UNet2DConditionModel with the MuseTalk-1.5 configuration (SD-1.x topology: in 8 / out 4,
    block_out_channels (320,640,1280,1280), layers_per_block 2, CrossAttnDown x3 + Down,
    Up + CrossAttnUp x3, 8 attention heads, cross_attention_dim 384, GroupNorm 32, SiLU,
    conv (not linear) proj_in/proj_out, GEGLU feed-forward, flip_sin_to_cos timestep embedding),
  * AutoencoderKL (sd-vae-ft-mse) decoder: post_quant_conv, conv_in, mid (resnet, 1-head
    attention, resnet), 4 up blocks x 3 resnets with nearest-2x upsample + conv, GN-SiLU-conv_out,
under diffusers' own state_dict key names, so a real checkpoint would load unchanged.

PARITY UNPINNED for the graph as a whole: no diffusers build, config file, checkpoint or golden tensor
exists in this container or in the reference tree to check this restatement against.  What IS pinned:
the ResnetBlock2D building block (GN -> SiLU -> conv -> GN -> SiLU -> conv + 1x1 shortcut) and the
asymmetric-pad stride-2 downsample, against the reference's own in-tree implementation
(avatars/musetalk/models/syncnet.py:71-139, imported with `diffusers` stubbed; golden tensors in
tests/golden/musetalk_blocks_golden.npz), the positional encoding (unet.py:12-27) and the Whisper side
(oracle/whisper_oracle.py calls the installed transformers); the Transformer2D composition (GN, 1x1 in, LN, attention +
residual, LN, GEGLU feed-forward + residual, 1x1 out + residual) against the reference's in-tree AttentionBlock2D
(syncnet.py:142-181, its two diffusers leaf classes replaced by the pinned multi-head attention and a literal GEGLU);
and the WHOLE VAE decoder / encoder graph against an independent implementation of the same network that is installed
here (transformers' Janus VQ-VAE = the taming / latent-diffusion autoencoder AutoencoderKL ports; tests/test_vae_pin.py,
2.7e-6).  Still unpinned: the U-Net's block wiring (down / mid / up order, skip concatenation, timestep embedding) and
diffusers' key names / config values.  The HIP path is required to match THIS statement within the fp16 tolerance
written in tests/test_musetalk_gpu.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ---------------------------------------------------------------------------------------------
# configuration (MuseTalk 1.5 musetalk.json as published upstream; sd-vae-ft-mse config.json)
# ---------------------------------------------------------------------------------------------
UNET_IN, UNET_OUT = 8, 4
UNET_CH = (320, 640, 1280, 1280)
UNET_HEADS = 8
UNET_CTX_DIM = 384
UNET_GROUPS = 32
UNET_EPS = 1e-5            # norm_eps (resnets, conv_norm_out)
ATTN_GN_EPS = 1e-6         # Transformer2DModel.norm
LN_EPS = 1e-5
TIME_DIM = 1280
DOWN_HAS_ATTN = (True, True, True, False)
UP_HAS_ATTN = (False, True, True, True)

VAE_CH = (128, 256, 512, 512)
VAE_LATENT = 4
VAE_GROUPS = 32
VAE_EPS = 1e-6
VAE_SCALING = 0.18215


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def _conv(sd: SD, p: str, x: Tensor, stride=1, pad=1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=pad)


# fp8 conv path (BASELINE.json configs[4]; include/ltk.h ltk_musetalk_set_fp8): emulation of what the engine does when it
# is enabled -- the SiLU(GroupNorm(x)) in front of every ResnetBlock2D 3x3 conv is multiplied by `ascale`, saturated to
# +-448 and rounded to OCP e4m3 (round to nearest even); the conv weights are rounded to e4m3 after a per-output-channel
# scale 224 / max|w|; products and sums are fp32.  Off by default: the reference has no fp8 mode, this is OUR statement
# of the quantised path, used to check the kernels, and its distance to the unquantised oracle is what the fp8 test reports.
FP8 = {"on": False, "ascale": 8.0}


def fp8_round(x: Tensor) -> Tensor:
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def fp8_weight(w: Tensor):
    """-> (dequantised weight, per-output-channel scale)"""
    m = w.abs().amax(dim=(1, 2, 3))
    # tensor / tensor: correctly rounded fp32 division like the engine's 224.f / m (torch evaluates `224.0 / m` as
    # 224 * reciprocal(m), one ulp off for some m, which flips the e4m3 rounding of weights that sit on a tie)
    sw = torch.where(m > 0, torch.full_like(m, 224.0) / m, torch.ones_like(m))
    return fp8_round(w * sw[:, None, None, None]) / sw[:, None, None, None], sw


def _conv3_q(sd: SD, p: str, x: Tensor) -> Tensor:
    """3x3 s1 p1 conv of a GroupNorm+SiLU output: fp8 operands when FP8 is on and the channel count allows it."""
    if not FP8["on"] or x.shape[1] % 32 != 0:
        return _conv(sd, p, x)
    a = float(FP8["ascale"])
    wq, _ = fp8_weight(sd[p + ".weight"])
    return F.conv2d(fp8_round(x * a) / a, wq, sd[p + ".bias"], padding=1)


def _linear(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd: SD, p: str, x: Tensor, groups: int, eps: float) -> Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def timestep_embedding(timesteps: Tensor, dim: int = 320) -> Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    return torch.cat([emb[:, half:], emb[:, :half]], dim=-1)   # cos first


def time_embed(sd: SD, timesteps: Tensor) -> Tensor:
    """UNet2DConditionModel.time_proj + time_embedding (linear -> SiLU -> linear) -> (1,1280)."""
    t = timestep_embedding(timesteps, UNET_CH[0])
    t = _linear(sd, "time_embedding.linear_1", t)
    t = F.silu(t)
    return _linear(sd, "time_embedding.linear_2", t)


def _tap(taps, name, t):
    if taps is not None:
        taps[name] = t.detach().clone()
    return t


def resnet(sd: SD, p: str, x: Tensor, temb: Optional[Tensor], groups: int, eps: float, taps=None) -> Tensor:
    """diffusers ResnetBlock2D (output_scale_factor 1, no up/down).  Without temb this is the reference's in-tree
    ResnetBlock2D (avatars/musetalk/models/syncnet.py:71-139): pinned against it by oracle/gen_golden_musetalk.py ->
    tests/golden/musetalk_blocks_golden.npz (tests/test_musetalk_host.py)."""
    h = _tap(taps, p + ".norm1", F.silu(_gn(sd, p + ".norm1", x, groups, eps)))
    h = _conv3_q(sd, p + ".conv1", h)
    if temb is not None:
        h = h + _linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    _tap(taps, p + ".conv1", h)
    h = _tap(taps, p + ".norm2", F.silu(_gn(sd, p + ".norm2", h, groups, eps)))
    h = _conv3_q(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = _tap(taps, p + ".conv_shortcut", _conv(sd, p + ".conv_shortcut", x, pad=0))
    return _tap(taps, p + ".conv2", x + h)


def downsample_asym(sd: SD, p: str, x: Tensor) -> Tensor:
    """diffusers Downsample2D(padding=0): one zero row / column at the bottom / right, then Conv2d(k3, s2, p0).
    Pinned (tests/golden/musetalk_blocks_golden.npz) against the in-tree equivalent, syncnet.py:112-121,136-138."""
    return _conv(sd, p, F.pad(x, (0, 1, 0, 1)), stride=2, pad=0)


def attention(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int, taps=None) -> Tensor:
    """diffusers Attention (AttnProcessor2_0): q/k/v projections, scaled dot-product, to_out.0."""
    B, T, C = x.shape
    q = _linear(sd, p + ".to_q", x)
    k = _linear(sd, p + ".to_k", ctx)
    v = _linear(sd, p + ".to_v", ctx)
    d = C // heads
    q = q.view(B, T, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)            # scale = d ** -0.5
    o = o.transpose(1, 2).reshape(B, T, C)
    _tap(taps, p + ".attn", o)            # (B, T, C) token-major
    return _linear(sd, p + ".to_out.0", o)


def transformer2d(sd: SD, p: str, x: Tensor, ctx: Optional[Tensor], taps=None, cross: bool = True, heads: int = UNET_HEADS,
                  groups: int = UNET_GROUPS, gn_eps: float = ATTN_GN_EPS) -> Tensor:
    """diffusers Transformer2DModel (use_linear_projection False) with one BasicTransformerBlock: GroupNorm -> 1x1 proj_in ->
    [LN -> self-attention + x; LN -> cross-attention + x; LN -> GEGLU feed-forward + x] -> 1x1 proj_out -> + input.
    `cross=False` drops the cross-attention sub-block: that is the reference's in-tree AttentionBlock2D
    (avatars/musetalk/models/syncnet.py:142-181), against which this function is pinned
    (oracle/gen_golden_musetalk.py, tests/golden/musetalk_blocks_golden.npz: same code path, one sub-block fewer).
    Token-major taps (B, T, C) are stored as (B, C, H, W) so they compare directly with the device tensors."""
    B, C, H, W = x.shape

    def tk(name, t):      # (B, T, c) -> (B, c, H, W)
        if taps is not None:
            taps[name] = t.detach().reshape(B, H, W, -1).permute(0, 3, 1, 2).clone()
        return t

    res = x
    h = _tap(taps, p + ".norm", _gn(sd, p + ".norm", x, groups, gn_eps))
    h = _tap(taps, p + ".proj_in", _conv(sd, p + ".proj_in", h, pad=0))
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = p + ".transformer_blocks.0"
    n = tk(b + ".norm1", _ln(sd, b + ".norm1", h))
    at = {} if taps is not None else None
    h = tk(b + ".attn1.to_out.0", attention(sd, b + ".attn1", n, n, heads, at) + h)
    if cross:
        n = tk(b + ".norm2", _ln(sd, b + ".norm2", h))
        h = tk(b + ".attn2.to_out.0", attention(sd, b + ".attn2", n, ctx, heads, at) + h)
    if at:
        for k, v in at.items():
            tk(k, v)
    n = tk(b + ".norm3", _ln(sd, b + ".norm3", h))
    g = tk(b + ".ff.net.0.proj", _linear(sd, b + ".ff.net.0.proj", n))               # GEGLU
    a, gate = g.chunk(2, dim=-1)
    h = tk(b + ".ff.net.2", _linear(sd, b + ".ff.net.2", tk(b + ".ff.geglu", a * F.gelu(gate))) + h)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = _conv(sd, p + ".proj_out", h, pad=0)
    return _tap(taps, p + ".proj_out", h + res)


def positional_encoding(x: Tensor) -> Tensor:
    """avatars/musetalk/models/unet.py:12-27 (d_model 384)."""
    b, t, d = x.shape
    pe = torch.zeros(t, d)
    position = torch.arange(0, t, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return x + pe[None].to(x.dtype)


# ---------------------------------------------------------------------------------------------
# UNet2DConditionModel.forward
# ---------------------------------------------------------------------------------------------
def unet_forward(sd: SD, latent: Tensor, ctx: Tensor, timestep: int = 0,
                 taps: Optional[Dict[str, Tensor]] = None, detail: Optional[str] = None) -> Tensor:
    """latent (B,8,32,32), ctx (B,50,384) (already position-encoded) -> (B,4,32,32)."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().clone()
        return t

    def dt(prefix):       # op-level taps only for blocks whose name starts with `detail`
        return taps if (taps is not None and detail is not None and prefix.startswith(detail)) else None

    temb = time_embed(sd, torch.tensor([timestep])).expand(latent.shape[0], -1)
    h = tap("conv_in", _conv(sd, "conv_in", latent))
    skips: List[Tensor] = [h]
    for i in range(4):
        for j in range(2):
            h = resnet(sd, f"down_blocks.{i}.resnets.{j}", h, temb, UNET_GROUPS, UNET_EPS, dt(f"down_blocks.{i}.resnets.{j}"))
            if DOWN_HAS_ATTN[i]:
                h = transformer2d(sd, f"down_blocks.{i}.attentions.{j}", h, ctx, dt(f"down_blocks.{i}.attentions.{j}"))
            skips.append(tap(f"down_blocks.{i}.{j}", h))
        if i < 3:
            h = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, pad=1)
            skips.append(tap(f"down_blocks.{i}.down", h))
    h = resnet(sd, "mid_block.resnets.0", h, temb, UNET_GROUPS, UNET_EPS)
    h = transformer2d(sd, "mid_block.attentions.0", h, ctx)
    h = tap("mid_block", resnet(sd, "mid_block.resnets.1", h, temb, UNET_GROUPS, UNET_EPS))
    for i in range(4):
        for j in range(3):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, f"up_blocks.{i}.resnets.{j}", h, temb, UNET_GROUPS, UNET_EPS, dt(f"up_blocks.{i}.resnets.{j}"))
            if UP_HAS_ATTN[i]:
                h = transformer2d(sd, f"up_blocks.{i}.attentions.{j}", h, ctx, dt(f"up_blocks.{i}.attentions.{j}"))
            tap(f"up_blocks.{i}.{j}", h)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = tap(f"up_blocks.{i}.up", _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h))
    h = F.silu(_gn(sd, "conv_norm_out", h, UNET_GROUPS, UNET_EPS))
    return tap("conv_out", _conv(sd, "conv_out", h))


# ---------------------------------------------------------------------------------------------
# AutoencoderKL.decode (sd-vae-ft-mse)
# ---------------------------------------------------------------------------------------------
def vae_attention(sd: SD, p: str, x: Tensor) -> Tensor:
    """diffusers Attention inside UNetMidBlock2D of the VAE: GroupNorm, 1 head, residual."""
    B, C, H, W = x.shape
    h = _gn(sd, p + ".group_norm", x, VAE_GROUPS, VAE_EPS)
    h = h.view(B, C, H * W).transpose(1, 2)
    h = attention(sd, p, h, h, 1)
    return h.transpose(1, 2).reshape(B, C, H, W) + x


def vae_decode(sd: SD, z: Tensor, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """AutoencoderKL.decode(z).sample: z (B,4,32,32) -> (B,3,256,256) RGB in ~[-1,1]."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().clone()
        return t

    h = _conv(sd, "post_quant_conv", z, pad=0)
    h = tap("decoder.conv_in", _conv(sd, "decoder.conv_in", h))
    h = resnet(sd, "decoder.mid_block.resnets.0", h, None, VAE_GROUPS, VAE_EPS)
    h = vae_attention(sd, "decoder.mid_block.attentions.0", h)
    h = tap("decoder.mid_block", resnet(sd, "decoder.mid_block.resnets.1", h, None, VAE_GROUPS, VAE_EPS))
    for i in range(4):
        for j in range(3):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, VAE_GROUPS, VAE_EPS)
        tap(f"decoder.up_blocks.{i}", h)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, VAE_GROUPS, VAE_EPS))
    return tap("decoder.conv_out", _conv(sd, "decoder.conv_out", h))


def vae_encode_moments(sd: SD, x: Tensor, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """AutoencoderKL.encode(x) up to the moments (mean | logvar): x (B,3,256,256) in [-1,1] -> (B,8,32,32).
    diffusers Encoder: conv_in, DownEncoderBlock2D x4 (2 resnets; Downsample2D(padding=0) = F.pad(0,1,0,1) + conv s2),
    UNetMidBlock2D, GN-SiLU-conv_out, then quant_conv."""
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(4):
        for j in range(2):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, VAE_GROUPS, VAE_EPS)
        if i < 3:
            h = downsample_asym(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h)
        if taps is not None:
            taps[f"encoder.down_blocks.{i}"] = h.detach().clone()
    h = resnet(sd, "encoder.mid_block.resnets.0", h, None, VAE_GROUPS, VAE_EPS)
    h = vae_attention(sd, "encoder.mid_block.attentions.0", h)
    h = resnet(sd, "encoder.mid_block.resnets.1", h, None, VAE_GROUPS, VAE_EPS)
    if taps is not None:
        taps["encoder.mid_block"] = h.detach().clone()
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, VAE_GROUPS, VAE_EPS))
    h = _conv(sd, "encoder.conv_out", h)
    return _conv(sd, "quant_conv", h, pad=0)


def get_latents_for_unet(vae_sd: SD, face_bgr, noise: Optional[Tensor] = None) -> Tensor:
    """avatars/musetalk/models/vae.py:55-94,110-122 for one 256x256 BGR array -> (1,8,32,32).
    noise (2,4,32,32): the standard-normal draws of latent_dist.sample() for (masked, reference); None -> the mean."""
    import numpy as np
    img = np.asarray(face_bgr)[..., ::-1]                             # cv2.cvtColor(BGR2RGB)
    x = np.asarray([img]) / 255.
    x = torch.squeeze(torch.FloatTensor(np.transpose(x, (3, 0, 1, 2))))
    mask = torch.zeros((256, 256))
    mask[:128, :] = 1
    outs = []
    for k, half_mask in enumerate((True, False)):
        xi = x * (mask > 0.5) if half_mask else x
        xi = ((xi - 0.5) / 0.5).unsqueeze(0)
        mom = vae_encode_moments(vae_sd, xi)
        mean, logvar = mom[:, :4], torch.clamp(mom[:, 4:], -30.0, 20.0)
        z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise[k][None]
        outs.append(VAE_SCALING * z)
    return torch.cat(outs, dim=1)


def decode_latents(vae_sd: SD, latents: Tensor):
    """avatars/musetalk/models/vae.py:96-108 -> uint8 (B,256,256,3) BGR."""
    image = vae_decode(vae_sd, (1 / VAE_SCALING) * latents)       # vae.py:102: the reciprocal, then a product
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.detach().cpu().permute(0, 2, 3, 1).float().numpy()
    image = (image * 255).round().astype("uint8")
    return image[..., ::-1]


def inference_batch(unet_sd: SD, vae_sd: SD, latent_list_cycle, index: int, batch_size: int, whisper_batch):
    """avatars/musetalk_avatar.py:130-152 on explicit weights: uint8 (B,256,256,3) BGR."""
    from .paste_oracle import mirror_index
    length = len(latent_list_cycle)
    latent = torch.cat([latent_list_cycle[mirror_index(length, index + i)] for i in range(batch_size)], dim=0)
    feat = positional_encoding(torch.as_tensor(whisper_batch, dtype=torch.float32))
    pred = unet_forward(unet_sd, latent.float(), feat)
    return decode_latents(vae_sd, pred)
