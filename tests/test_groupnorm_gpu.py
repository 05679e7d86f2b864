"""The three GroupNorm(+SiLU) kernels of the MuseTalk path against a plain PyTorch fp32 reference of the same op
(torch.nn.functional.group_norm + silu: what diffusers' ResnetBlock2D.norm1/norm2, Transformer2DModel.norm and the decoder's
conv_norm_out compute; call sites avatars/musetalk/models/unet.py:36-46, vae.py:96-108), through ltk_groupnorm_f16:

  impl 1  gn_stats_kernel + gn_apply_kernel   two launches, three tensor passes, any shape
  impl 2  gn_group_kernel                     one block per (image, group): the U-Net maps, the VAE's 32^2 maps
  impl 3  gn_coop_kernel                      one tensor pass, the blocks of an (image, 16-channel block) exchange partial sums:
                                              the VAE's 64^2 .. 256^2 maps (2 .. 32 members per set, 4 / 8 / 16 channels per group)

Tolerance: fp16 output rounding (2^-11 relative) on top of fp32 statistics: |err| <= 2e-3 * max(|ref|, 1) for fp16 outputs; for the e4m3
outputs of the fp8 conv path half an e4m3 step (2^-4 relative) + the subnormal step."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from livetalking_amd.layout import empty_cb16, from_cb16, to_cb16  # noqa: E402

# (N, C, H, W, groups, silu, impls that must serve the shape; the others are tried too and may refuse it)
CASES = [
    (3, 128, 256, 256, 32, True, (1, 3)),      # 32 members per set, 4 channels per group
    (2, 256, 256, 256, 32, True, (1, 3)),      # 8 channels per group
    (2, 256, 128, 128, 32, True, (1, 3)),      # 8 members
    (3, 512, 128, 128, 32, False, (1, 3)),     # 16 channels per group = one group per channel block
    (5, 512, 64, 64, 32, True, (1, 3)),        # 2 members
    (1, 128, 64, 64, 32, True, (1, 3)),        # one image
    (2, 512, 32, 32, 32, True, (1, 2)),        # the VAE's 32^2 maps: one block per (image, group)
    (4, 320, 32, 32, 32, True, (1, 2)),        # U-Net: 10 channels per group (groups straddle 16-channel blocks)
    (3, 640, 16, 16, 32, True, (1, 2)),
    (2, 1280, 8, 8, 32, False, (1, 2)),
    (2, 2560, 4, 4, 32, True, (1, 2)),
    (2, 128, 60, 50, 32, True, (1,)),          # ragged pixel count: two-pass kernels only
]


def _ref(x, groups, gamma, beta, silu, eps):
    y = F.group_norm(x, groups, gamma, beta, eps)
    return F.silu(y) if silu else y


@pytest.mark.gpu
def test_groupnorm_kernels_vs_torch(engine):
    report = []
    for i, (N, C, H, W, groups, silu, impls) in enumerate(CASES):
        g = torch.Generator(device="cpu").manual_seed(500 + i)
        # per-channel offsets and gains so that group statistics matter; a few large values
        x = torch.randn(N, C, H, W, generator=g) * (0.5 + torch.rand(1, C, 1, 1, generator=g) * 3.0) + torch.randn(1, C, 1, 1, generator=g) * 2.0
        x = x.half().float()
        gamma = torch.rand(C, generator=g) + 0.5
        beta = torch.randn(C, generator=g) * 0.3
        eps = 1e-6 if i % 2 else 1e-5
        ref = _ref(x.cuda(), groups, gamma.cuda(), beta.cuda(), silu, eps)
        x_dev = to_cb16(x.cuda())
        outs = {}
        for impl in (0, 1, 2, 3):
            y = empty_cb16(N, C, H, W, fill=float("nan"))
            try:
                engine.groupnorm_f16(x_dev.data_ptr(), N, C, H * W, groups, eps, gamma.numpy(), beta.numpy(), silu, y.data_ptr(), impl=impl)
            except RuntimeError:
                assert impl not in impls, f"case {i}: impl {impl} refused a shape it is expected to serve"
                continue
            torch.cuda.synchronize()
            got = from_cb16(y, C)
            err = (got - ref).abs()
            bad = int((~(err <= 2e-3 * ref.abs().clamp(min=1.0))).sum())
            outs[impl] = got
            print(f"[gn] case {i} N={N} C={C} {H}x{W} silu={silu} impl={impl}: bad={bad} max err {float(torch.nan_to_num(err, nan=1e9).max()):.3g}")
            if bad:
                report.append(f"case {i} impl {impl}: {bad} values out of tolerance")
        # the implementations agree within one fp16 step of each other, and the program's own choice is one of them
        served = [k for k in outs if k != 0]
        base = outs[1]
        for impl in served[1:]:
            d = (outs[impl] - base).abs()
            if not bool((d <= 2e-3 * base.abs().clamp(min=1.0)).all()):
                report.append(f"case {i}: impl {impl} differs from impl 1 by {float(d.max()):.3g}")
        assert any(torch.equal(outs[0], outs[k]) for k in served), f"case {i}: impl 0 is none of {served}"
    assert not report, "\n".join(report)


@pytest.mark.gpu
def test_groupnorm_refuses_shapes_a_kernel_does_not_serve(engine):
    x = to_cb16(torch.randn(1, 128, 100, 100).cuda())
    y = empty_cb16(1, 128, 100, 100)
    ones, zeros = np.ones(128, np.float32), np.zeros(128, np.float32)
    for impl in (2, 3):        # 10 000 pixels x 2 channel pairs exceed one block's registers; 10 000 is no multiple of 2048
        with pytest.raises(RuntimeError):
            engine.groupnorm_f16(x.data_ptr(), 1, 128, 10000, 32, 1e-6, ones, zeros, True, y.data_ptr(), impl=impl)
    engine.groupnorm_f16(x.data_ptr(), 1, 128, 10000, 32, 1e-6, ones, zeros, True, y.data_ptr(), impl=0)       # the two-pass kernels take it


@pytest.mark.gpu
def test_groupnorm_e4m3_output_and_determinism(engine):
    """The writers of the fp8 conv path's operands (e4m3 [N][C/32][P][32] = saturate(result * scale)) against the same op in torch, for the
    one-pass and the two-pass kernels; and the cooperative kernel run ten times gives the same bytes (fixed summation order)."""
    N, C, H, W, groups, scale = 3, 256, 128, 128, 32, 8.0
    g = torch.Generator(device="cpu").manual_seed(77)
    x = (torch.randn(N, C, H, W, generator=g) * 2.0 + torch.randn(1, C, 1, 1, generator=g)).half().float()
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    ref = (_ref(x.cuda(), groups, gamma.cuda(), beta.cuda(), True, 1e-6) * scale).clamp(-448.0, 448.0)
    x_dev = to_cb16(x.cuda())
    first = None
    for impl in (1, 3):
        y = torch.zeros(N, C // 32, H, W, 32, dtype=torch.uint8, device="cuda")
        engine.groupnorm_f16(x_dev.data_ptr(), N, C, H * W, groups, 1e-6, gamma.numpy(), beta.numpy(), True, y.data_ptr(), impl=impl, out_fp8=True,
                             out_scale=scale)
        torch.cuda.synchronize()
        got = y.view(torch.float8_e4m3fn).float().permute(0, 1, 4, 2, 3).reshape(N, C, H, W)
        err = (got - ref).abs()
        tol = ref.abs() * 2.0 ** -4 + 2.0 ** -9 + 1e-3 * ref.abs()
        bad = int((~(err <= tol)).sum())
        print(f"[gn e4m3] impl {impl}: bad={bad} of {err.numel()}, max err {float(err.max()):.3g}")
        assert bad == 0
    y16 = []
    for _ in range(10):
        y = empty_cb16(N, C, H, W, fill=float("nan"))
        engine.groupnorm_f16(x_dev.data_ptr(), N, C, H * W, groups, 1e-6, gamma.numpy(), beta.numpy(), True, y.data_ptr(), impl=3)
        torch.cuda.synchronize()
        y16.append(y.clone())
        first = y16[0]
        assert torch.equal(y16[-1].view(torch.int16), first.view(torch.int16)), "the cooperative GroupNorm is not deterministic"
