"""Precision study (CPU, TEST INFRASTRUCTURE ONLY - see oracle/__init__.py; nothing in the product path imports this): would Winograd F(2x2, 3x3) on fp16
MFMA operands hold the engine's parity bar on the Wav2Lip generator?

The engine computes every conv as fp16 operands x fp32 accumulation with fp16 activations between layers (60.6 dB / max 1 LSB against the
reference's fp32 frames).  F(2x2, 3x3) needs 4 multiplies per output instead of 9 (2.25x fewer MFMA cycles on the 3x3 stride-1 layers = 80 % of the
generator's MACs) but feeds the matrix unit TRANSFORMED operands: U = G g G^T (weights, transformed in fp32 at load, then rounded to fp16) and
V = B^T d B (input tiles, transformed from fp16 activations; sums of up to four values, rounded to fp16 again).  This script runs the oracle's layer walk
(oracle/wav2lip_oracle.py, i.e. wav2lip_v2.py:123-163) three ways on the golden fixture's inputs -

  fp32          the oracle itself (sanity: reproduces the golden frames)
  fp16-direct   what the engine does today: fp16 weights and activations, fp32 accumulation
  fp16-wino     the same, with every 3x3 stride-1 pad-1 conv on F(2x2, 3x3): V and U rounded to fp16, products accumulated in fp32,
                the output transform A^T M A in fp32 (variants: V computed in fp32 and rounded once / in fp16 arithmetic, rounded after each 1-D pass)

- and prints PSNR / max LSB of the uint8 frames against the reference's golden frames, and the per-layer relative error of the 3x3 layers.

    python -m oracle.study_winograd            (about a minute on 8 cores)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from oracle import plugin_oracle, wav2lip_oracle as wo  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def h(x):
    return x.half().float()


def wino_conv(x, w, v_mode):
    """3x3 stride-1 pad-1 conv of fp16-valued x (B,C,H,W) with fp32 weights w (O,C,3,3) by F(2x2,3x3); H, W even."""
    B, C, H, W = x.shape
    U = h(torch.einsum("ai,ocij,bj->ocab", G, w, G))                                  # fp32 transform, one rounding
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                              # (B,C,H/2,W/2,4,4)
    if v_mode == "fp32":
        V = h(torch.einsum("ai,bcyxij,ej->bcyxae", BT, d, BT))
    else:                                                                               # fp16 arithmetic: a rounding after each 1-D pass
        V = h(torch.einsum("ej,bcyxaj->bcyxae", BT, h(torch.einsum("ai,bcyxij->bcyxaj", BT, d))))
    M = torch.einsum("ocae,bcyxae->boyxae", U, V)                                       # fp32 accumulation over input channels
    Y = torch.einsum("pa,boyxae,qe->boyxpq", AT, M, AT)                                 # (B,O,H/2,W/2,2,2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def block(x, sd, l, mode):
    w, b = sd[l.prefix + ".conv_block.0.weight"], sd[l.prefix + ".conv_block.0.bias"]
    wino = mode.startswith("wino") and l.kind == "conv" and l.k == (3, 3) and l.stride == (1, 1) and l.pad == (1, 1) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
    if mode == "fp32":
        return wo._block(x, sd, l), False
    if wino:
        y = wino_conv(x, w, "fp32" if mode == "wino" else "fp16") + b.view(1, -1, 1, 1)
    elif mode in ("fold", "fold1") and l.kind == "conv":
        # BN scale folded into the weights BEFORE their fp16 rounding (the accumulator then carries conv * scale, and a residual can be added to it
        # as x * 1.0 by one more MFMA on the staged centre pixel - exact - instead of being read again in the epilogue)
        sc = sd[l.prefix + ".conv_block.1.weight"] / torch.sqrt(sd[l.prefix + ".conv_block.1.running_var"] + wo.BN_EPS)
        sh = sd[l.prefix + ".conv_block.1.bias"] + (b - sd[l.prefix + ".conv_block.1.running_mean"]) * sc
        ws = w * sc.view(-1, 1, 1, 1)
        if mode == "fold1" and l.residual:
            # ... or with NO extra MFMA: the identity added to the centre tap of the scaled weights before their fp16 rounding (values near 1.0 there:
            # absolute rounding error up to 2^-11 on those Cout weights instead of a relative one)
            ws = ws.clone()
            i = torch.arange(l.cout)
            ws[i, i, l.k[0] // 2, l.k[1] // 2] += 1.0
        y = F.conv2d(x, h(ws), None, stride=l.stride, padding=l.pad) + sh.view(1, -1, 1, 1)
        if l.residual and mode == "fold":
            y = y + x
        return h(F.relu(y).clamp(max=65504.)), False
    elif l.kind == "conv":
        y = F.conv2d(x, h(w), b, stride=l.stride, padding=l.pad)
    else:
        y = F.conv_transpose2d(x, h(w), b, stride=l.stride, padding=l.pad, output_padding=l.out_pad)
    y = F.batch_norm(y, sd[l.prefix + ".conv_block.1.running_mean"], sd[l.prefix + ".conv_block.1.running_var"], sd[l.prefix + ".conv_block.1.weight"],
                     sd[l.prefix + ".conv_block.1.bias"], training=False, eps=wo.BN_EPS)
    if l.residual:
        y = y + x
    return h(F.relu(y).clamp(max=65504.)), wino


@torch.no_grad()
def forward(sd, mel, face, mode, taps):
    nw = [0, 0]
    def run(x, l):
        y, wino = block(x, sd, l, mode)
        taps[l.prefix] = y
        nw[0] += wino; nw[1] += 1
        return y
    x = mel if mode == "fp32" else h(mel)
    for l in wo.AUDIO_ENCODER:
        x = run(x, l)
    emb, feats = x, []
    x = face if mode == "fp32" else h(face)
    for blk in wo.FACE_ENCODER_BLOCKS:
        for l in blk:
            x = run(x, l)
        feats.append(x)
    x = emb
    for blk in wo.FACE_DECODER_BLOCKS:
        for l in blk:
            x = run(x, l)
        x = torch.cat((x, feats.pop()), dim=1)
    # output block + head: the engine keeps the 32-channel map in fp32 registers (fused head, conv3_head_kernel)
    l = wo.OUTPUT_CONV
    if mode == "fp32":
        x = wo._block(x, sd, l)
    else:
        w, b = sd[l.prefix + ".conv_block.0.weight"], sd[l.prefix + ".conv_block.0.bias"]
        y = (wino_conv(x, w, "fp32" if mode == "wino" else "fp16") if mode.startswith("wino") else F.conv2d(x, h(w), None, padding=1)) + b.view(1, -1, 1, 1)   # (fold: as direct)
        y = F.batch_norm(y, sd[l.prefix + ".conv_block.1.running_mean"], sd[l.prefix + ".conv_block.1.running_var"], sd[l.prefix + ".conv_block.1.weight"],
                         sd[l.prefix + ".conv_block.1.bias"], training=False, eps=wo.BN_EPS)
        x = F.relu(y)
        nw[0] += mode.startswith("wino"); nw[1] += 1
    x = F.conv2d(x, sd[wo.OUTPUT_HEAD_PREFIX + ".weight"], sd[wo.OUTPUT_HEAD_PREFIX + ".bias"])
    return torch.sigmoid(x), nw


def main():
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    g, gm = np.load(os.path.join(gd, "wav2lip_golden.npz")), np.load(os.path.join(gd, "mel_golden.npz"))
    sd = {k: torch.from_numpy(v) for k, v in synth.wav2lip_state_dict(int(g["weight_seed"])).items()}
    _, faces, _ = synth.wav2lip_avatar(int(g["avatar_frames"]), tuple(int(v) for v in g["avatar_hw"]), int(g["avatar_box"]), int(g["avatar_seed"]))
    B, index = int(g["batch"]), int(g["index"])
    mel_t, img_t = plugin_oracle.pack_inputs(faces, index, B, [gm["ref_chunks"][int(g["mel_step"])][i] for i in range(B)])
    ref = g["ref_pred_u8"].astype(np.int32)
    taps = {}
    for mode in ("fp32", "direct", "fold", "fold1", "wino", "wino16"):
        taps[mode] = {}
        pred, nw = forward(sd, mel_t, img_t, mode, taps[mode])
        u8 = (pred.numpy().transpose(0, 2, 3, 1) * np.float32(255.)).astype(np.uint8).astype(np.int32)
        d = np.abs(u8 - ref)
        mse = float((d.astype(np.float64) ** 2).mean())
        psnr = 10 * np.log10(255. ** 2 / mse) if mse > 0 else float("inf")
        print(f"{mode:8s} layers on Winograd {nw[0]:2d}/{nw[1]}   frames vs reference golden: PSNR {psnr:6.2f} dB, max {d.max()} LSB, "
              f"{100 * (d > 0).mean():.3f} % of bytes differ, {100 * (d > 1).mean():.4f} % by more than 1", flush=True)
    print("\nper-layer relative L2 error against the fp32 walk (3x3 stride-1 layers):   direct     fold    fold1     wino   wino16")
    for l in wo.all_block_layers():
        if not (l.kind == "conv" and l.k == (3, 3) and l.stride == (1, 1) and l.pad == (1, 1)) or l.prefix not in taps["fp32"]:
            continue
        r = taps["fp32"][l.prefix]
        e = [float((taps[m][l.prefix] - r).norm() / (r.norm() + 1e-30)) for m in ("direct", "fold", "fold1", "wino", "wino16")]
        print(f"  {l.prefix:28s} {tuple(r.shape[1:])!s:18s} {e[0]:.2e} {e[1]:.2e} {e[2]:.2e} {e[3]:.2e} {e[4]:.2e}")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    main()
