"""HIP implicit-GEMM conv kernel vs a plain PyTorch fp32 reference of the same op
(every distinct layer geometry of avatars/wav2lip/models/wav2lip_v2.py:12-91)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from livetalking_amd.layout import empty_cb16, from_cb16, to_cb16  # noqa: E402

# (N, H, W, Cin, Cout, k, stride, pad, transposed, out_pad, residual)
CASES = [
    # audio encoder
    (3, 80, 16, 1, 32, 3, 1, 1, False, 0, False),
    (3, 80, 16, 32, 32, 3, 1, 1, False, 0, True),
    (3, 80, 16, 32, 64, 3, (3, 1), 1, False, 0, False),
    (3, 27, 16, 64, 64, 3, 1, 1, False, 0, True),
    (3, 27, 16, 64, 128, 3, 3, 1, False, 0, False),
    (3, 9, 6, 128, 128, 3, 1, 1, False, 0, True),
    (3, 9, 6, 128, 256, 3, (3, 2), 1, False, 0, False),
    (3, 3, 3, 256, 256, 3, 1, 1, False, 0, True),
    (3, 3, 3, 256, 512, 3, 1, 0, False, 0, False),
    (3, 1, 1, 512, 512, 1, 1, 0, False, 0, False),
    (19, 1, 1, 512, 512, 1, 1, 0, False, 0, False),
    # face encoder
    (2, 256, 256, 6, 16, 7, 1, 3, False, 0, False),
    (2, 256, 256, 16, 32, 3, 2, 1, False, 0, False),
    (2, 128, 128, 32, 32, 3, 1, 1, False, 0, True),
    (2, 128, 128, 32, 64, 3, 2, 1, False, 0, False),
    (2, 64, 64, 64, 64, 3, 1, 1, False, 0, True),
    (2, 64, 64, 64, 128, 3, 2, 1, False, 0, False),
    (2, 32, 32, 128, 128, 3, 1, 1, False, 0, True),
    (2, 32, 32, 128, 256, 3, 2, 1, False, 0, False),
    (3, 16, 16, 256, 256, 3, 1, 1, False, 0, True),
    (3, 16, 16, 256, 512, 3, 2, 1, False, 0, False),
    (3, 8, 8, 512, 512, 3, 1, 1, False, 0, True),
    (3, 8, 8, 512, 512, 3, 2, 1, False, 0, False),
    (5, 4, 4, 512, 512, 3, 1, 1, False, 0, True),
    (5, 4, 4, 512, 512, 4, 1, 0, False, 0, False),
    # decoder
    (3, 1, 1, 1024, 512, 4, 1, 0, True, 0, False),
    (3, 4, 4, 1024, 512, 3, 2, 1, True, 1, False),
    (3, 8, 8, 1024, 512, 3, 2, 1, True, 1, False),
    (2, 16, 16, 768, 384, 3, 2, 1, True, 1, False),
    (2, 32, 32, 384, 384, 3, 1, 1, False, 0, True),
    (2, 32, 32, 512, 256, 3, 2, 1, True, 1, False),
    (2, 64, 64, 256, 256, 3, 1, 1, False, 0, True),
    (2, 64, 64, 320, 128, 3, 2, 1, True, 1, False),
    (1, 128, 128, 160, 64, 3, 2, 1, True, 1, False),
    (1, 256, 256, 64, 64, 3, 1, 1, False, 0, True),
    (1, 256, 256, 80, 32, 3, 1, 1, False, 0, False),
    # ragged maps (partial tiles in both directions)
    (2, 37, 21, 32, 32, 3, 1, 1, False, 0, True),
    (2, 45, 70, 64, 64, 3, 2, 1, False, 0, False),
]


def _pair(v):
    return (v, v) if isinstance(v, int) else v


def run_case(eng, case, seed, v3=1):
    from livetalking_amd.engine import Engine
    Engine.set_knob("CONV_V3", v3)        # read when the layer plan is created
    N, H, W, Cin, Cout, k, stride, pad, transposed, out_pad, residual = case
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g).half().float()
    fan = Cin * k * k / (_pair(stride)[0] * _pair(stride)[1] if transposed else 1)
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = (torch.randn(wshape, generator=g) * (2.0 / fan) ** 0.5).half().float()
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    xd, wd = x.cuda(), w.cuda()
    if transposed:
        ref = F.conv_transpose2d(xd, wd, None, stride=_pair(stride), padding=_pair(pad), output_padding=out_pad)
    else:
        ref = F.conv2d(xd, wd, None, stride=_pair(stride), padding=_pair(pad))
    ref = ref * scale.cuda()[None, :, None, None] + shift.cuda()[None, :, None, None]
    x_dev = to_cb16(xd)                      # engine layout: [N][C/16][H][W][16] ([N][H][W][8] for C <= 8)
    res_ptr = 0
    if residual:
        ref = ref + xd
        res_ptr = x_dev.data_ptr()
    ref = torch.relu(ref)
    Ho, Wo = ref.shape[2], ref.shape[3]
    y = empty_cb16(N, Cout, Ho, Wo, fill=float("nan"))
    eng.conv2d_f16(x_dev.data_ptr(), N, H, W, Cin, w.numpy(), Cout, k, stride, pad, transposed, out_pad,
                   scale.numpy(), shift.numpy(), res_ptr, True, y.data_ptr())
    torch.cuda.synchronize()
    got = from_cb16(y, Cout)
    err = (got - ref).abs()
    tol = 2e-3 * ref.abs().clamp(min=1.0) + 2e-3
    bad = (~(err <= tol)).sum().item()   # NaNs count as bad
    return bad, float(torch.nan_to_num(err, nan=1e9).max()), float(ref.abs().max())


# conv3 extras: batch sizes that exercise the 512-pixel tiles, the split-K path (small maps, deep K) with and
# without residual, partial cout tiles and ragged maps
CASES_V3 = [
    (16, 64, 64, 64, 64, 3, 1, 1, False, 0, True),
    (16, 32, 32, 128, 128, 3, 1, 1, False, 0, True),
    (16, 8, 8, 512, 512, 3, 1, 1, False, 0, True),
    (16, 4, 4, 512, 512, 3, 1, 1, False, 0, True),
    (16, 16, 16, 256, 256, 3, 1, 1, False, 0, True),
    (16, 1, 1, 512, 512, 1, 1, 0, False, 0, False),
    (16, 1, 1, 1024, 512, 4, 1, 0, True, 0, False),
    (16, 4, 4, 1024, 512, 3, 2, 1, True, 1, False),
    (16, 8, 8, 1024, 512, 3, 2, 1, True, 1, False),
    (7, 64, 64, 320, 128, 3, 2, 1, True, 1, False),
    (5, 128, 128, 160, 64, 3, 2, 1, True, 1, False),
    (4, 256, 256, 80, 32, 3, 1, 1, False, 0, False),
    (3, 37, 21, 64, 96, 3, 1, 1, False, 0, False),
    (3, 19, 45, 96, 48, 3, 2, 1, True, 1, False),
    (2, 33, 17, 128, 80, 1, 1, 0, False, 0, False),
]


def _run_all(engine, cases, v3):
    report = []
    for i, case in enumerate(cases):
        try:
            bad, maxerr, refmax = run_case(engine, case, 100 + i, v3)
        except Exception as ex:  # report all cases, not just the first
            report.append(f"case {i} {case}: EXC {ex}")
            continue
        status = "ok" if bad == 0 else "FAIL"
        print(f"[conv v3={v3}] {status} case {i} {case}: bad={bad} maxerr={maxerr:.4g} refmax={refmax:.3g}")
        if bad:
            report.append(f"case {i} {case}: {bad} elements out of tolerance, max err {maxerr:.4g} (ref max {refmax:.3g})")
    return report


@pytest.mark.gpu
def test_conv_every_wav2lip_geometry(engine):
    """Default kernel selection (conv3 where eligible, conv_mfma for the 7x7 / strided layers)."""
    report = _run_all(engine, CASES + CASES_V3, 1)
    assert not report, "\n".join(report)


@pytest.mark.gpu
def test_conv_first_generation_kernel(engine):
    """conv_mfma.hip alone (LTK_CONV_V3=0) stays correct: it is the fallback for every geometry."""
    try:
        report = _run_all(engine, CASES, 0)
    finally:
        from livetalking_amd.engine import Engine
        Engine.set_knob("CONV_V3", 1)
    assert not report, "\n".join(report)


# lin_fk_kernel (conv3_mfma.hip): 1x1 / linear layers with K = 320 / 384 / 512 / 640 / 1280 on >= 512 rows (MuseTalk's 32^2 / 16^2 transformer levels):
# whole and ragged pixel tiles, tiles that straddle images (8x8 maps), partial cout slabs (48 = 1.5 slabs), with / without residual
CASES_LIN_FK = [
    (16, 32, 32, 320, 320, 1, 1, 0, False, 0, True),
    (16, 32, 32, 320, 960, 1, 1, 0, False, 0, False),
    (16, 16, 16, 640, 640, 1, 1, 0, False, 0, True),
    (3, 32, 32, 640, 320, 1, 1, 0, False, 0, False),
    (5, 21, 37, 320, 48, 1, 1, 0, False, 0, False),
    (40, 8, 8, 640, 640, 1, 1, 0, False, 0, True),
    (33, 8, 8, 320, 336, 1, 1, 0, False, 0, False),
    (2, 32, 32, 512, 512, 1, 1, 0, False, 0, True),
    (13, 10, 5, 384, 160, 1, 1, 0, False, 0, False),
    (1, 32, 32, 320, 320, 1, 1, 0, False, 0, True),
    # K = 1280: two waves per pixel subtile (K halves summed through LDS: fixed order, not conv3's)
    (16, 8, 8, 1280, 1280, 1, 1, 0, False, 0, True),
    (3, 32, 32, 1280, 320, 1, 1, 0, False, 0, False),
    (9, 8, 8, 1280, 3840, 1, 1, 0, False, 0, False),
    (11, 7, 9, 1280, 48, 1, 1, 0, False, 0, False),
    # K = 2560 / 5120: lin_mp_kernel (passes of 1280 channels, the accumulators of LIN_MP weight slabs per block)
    (16, 16, 16, 2560, 640, 1, 1, 0, False, 0, False),
    (16, 8, 8, 5120, 1280, 1, 1, 0, False, 0, False),
    (9, 8, 8, 2560, 1280, 1, 1, 0, False, 0, False),
    (11, 7, 9, 2560, 48, 1, 1, 0, False, 0, False),
    (5, 11, 13, 5120, 336, 1, 1, 0, False, 0, False),
    (2, 16, 16, 2560, 2560, 1, 1, 0, False, 0, True),
]


@pytest.mark.gpu
def test_lin_fk_kernel_vs_torch_and_conv3(engine):
    """The short-K linear kernel against the fp32 torch reference, and bit for bit against conv3's 1x1 path with the split-K
    rule off (both sum the channel blocks in ascending order)."""
    from livetalking_amd.engine import Engine

    def run(case, seed):
        N, H, W, Cin, Cout, k, stride, pad, transposed, out_pad, residual = case
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.randn(N, Cin, H, W, generator=g).half().float()
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).half().float()
        scale = torch.rand(Cout, generator=g) + 0.5
        shift = torch.randn(Cout, generator=g) * 0.1
        x_dev = to_cb16(x.cuda())
        y = empty_cb16(N, Cout, H, W, fill=float("nan"))
        engine.conv2d_f16(x_dev.data_ptr(), N, H, W, Cin, w.numpy(), Cout, 1, 1, 0, False, 0, scale.numpy(), shift.numpy(),
                          x_dev.data_ptr() if residual else 0, True, y.data_ptr())
        torch.cuda.synchronize()
        ref = F.conv2d(x.cuda(), w.cuda()) * scale.cuda()[None, :, None, None] + shift.cuda()[None, :, None, None]
        if residual:
            ref = ref + x.cuda()
        return y.clone(), torch.relu(ref)

    report = []
    try:
        Engine.set_knob("SPLITK", 0)
        runs = [(i, case, mp) for i, case in enumerate(CASES_LIN_FK) for mp in ((2, 3) if case[3] >= 2560 else (1,))]   # lin_mp: both slab counts, forced
        for i, case, mp in runs:
            Engine.set_knob("LIN_FK", 1)
            Engine.set_knob("LIN_MP", mp)
            y1, ref = run(case, 300 + i)
            Engine.set_knob("LIN_FK", 0)
            y0, _ = run(case, 300 + i)
            Cout = case[4]
            got = from_cb16(y1, Cout)
            err = (got - ref).abs()
            bad = (~(err <= 2e-3 * ref.abs().clamp(min=1.0) + 2e-3)).sum().item()
            same = torch.equal(from_cb16(y1, Cout), from_cb16(y0, Cout))
            dd = (from_cb16(y1, Cout) - from_cb16(y0, Cout)).abs()
            print(f"[lin_fk] case {i} {case} LIN_MP={mp}: bad={bad} maxerr={float(torch.nan_to_num(err, nan=1e9).max()):.4g} equal_to_conv3={same} "
                  f"(differing {int((dd != 0).sum())} of {dd.numel()}, max {float(dd.max()):.3g})")
            if bad or (not same and case[3] < 1280) or float(dd.max()) > 4e-3 * float(ref.abs().max()):
                report.append(f"case {i} {case}: bad={bad} equal_to_conv3={same}")
    finally:
        Engine.set_knob("LIN_FK", 1)
        Engine.set_knob("LIN_MP", 1)
        Engine.set_knob("SPLITK", 1)
    assert not report, "\n".join(report)
