"""-m gpu: every C-ABI entry point, run on the MI355X through ctypes, against the numpy restatement of the same
contract (tests/abi_emulator.py) on identical inputs.  fp32 results must agree to ~1e-5 relative (the fp32 MFMA path
is exact per product; only the summation order differs); bf16 results to 1 bf16 ulp-ish (2^-7 relative to the tensor
maximum after identical rounding of the stored outputs).  Integer / mask outputs are bit-exact."""
import ctypes
import math

import numpy as np
import pytest
import torch

from ddpm_torch import _hip
from tests.abi_emulator import Emulator

pytestmark = pytest.mark.gpu
DT = {0: torch.float32, 1: torch.bfloat16}
TOL = {0: 2e-5, 1: 1.2e-2}
DEV = "cuda:0"


class A:
    """tensor argument: host master copy `t`, element offset, and whether the kernel writes it."""

    def __init__(self, t, out=False, off=0, name=""):
        self.t, self.out, self.off, self.name = t.contiguous(), out, off, name


def both(name, *args, tol=None, atol=0.0):
    emu = Emulator()
    hp, dp, outs, keep = [], [], [], []
    for a in args:
        if isinstance(a, A):
            d = a.t.cuda()
            keep.append(d)                        # device copies must outlive the launch (caching allocator reuse)
            hp.append(a.t.data_ptr() + a.off * a.t.element_size())
            dp.append(d.data_ptr() + a.off * a.t.element_size())
            if a.out:
                outs.append((a, d))
        elif a is None:
            hp.append(0); dp.append(0)
        else:
            hp.append(a); dp.append(a)
    _hip.call(name, *dp, _hip.stream())
    torch.cuda.synchronize()
    emu.call(name, *hp, 0)
    worst = 0.0
    for a, d in outs:
        ref, got = a.t.float(), d.cpu().float()
        assert torch.isfinite(got).all(), f"{name}:{a.name} not finite"
        scale = float(ref.abs().max()) or 1.0
        err = float((ref - got).abs().max())
        rt = tol if tol is not None else 2e-5
        assert err <= rt * scale + atol, f"{name}:{a.name} max err {err:.3e} (scale {scale:.3e}, tol {rt:.1e})"
        worst = max(worst, err / scale)
    return worst


def r(*shape, seed, dt=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DT[dt])


CONV_CASES = [
    # B, H, W, C, N, R, stride, pad_t, pad_l, ups, dil, Ho, Wo
    ("3x3", 2, 8, 8, 32, 64, 3, 1, 1, 1, 0, 0, 8, 8),
    ("3x3_multi_tile", 3, 16, 16, 128, 256, 3, 1, 1, 1, 0, 0, 16, 16),
    ("1x1", 2, 4, 4, 64, 96, 1, 1, 0, 0, 0, 0, 4, 4),
    ("s2_even", 2, 8, 8, 32, 32, 3, 2, 0, 0, 0, 0, 4, 4),
    ("s2_odd", 2, 9, 9, 32, 32, 3, 2, 1, 1, 0, 0, 5, 5),
    ("upsample", 2, 4, 4, 32, 32, 3, 1, 1, 1, 1, 0, 8, 8),
    ("dgrad_s2", 2, 4, 4, 32, 32, 3, 1, 2, 2, 0, 1, 8, 8),
    ("ktail_c8", 2, 8, 8, 8, 128, 3, 1, 1, 1, 0, 0, 8, 8),
    ("k4608", 1, 4, 4, 512, 256, 3, 1, 1, 1, 0, 0, 4, 4),
]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd(case, dt):
    _, B, H, W, C, N, R, stride, pt, pl, ups, dil, Ho, Wo = case
    ld, yld = C + 16, N + 32                                   # exercise pitches
    x = r(B * H * W, ld, seed=1, dt=dt)
    w = r(N, R * R * C, seed=2, dt=dt, scale=1.0 / math.sqrt(R * R * C))
    bias, rowb = r(N, seed=3), r(B, N + 8, seed=4)
    res = r(B * Ho * Wo, yld, seed=5, dt=dt)
    y = r(B * Ho * Wo, yld, seed=6, dt=dt)
    for acc in (0, 1):
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y"), yld, A(bias), A(rowb), N + 8, A(res), yld,
             B, H, W, C, Ho, Wo, N, R, R, stride, pt, pl, ups, dil, acc, 0, 1, None, None, dt, tol=TOL[dt])
    # fp32 output modes (NHWC fp32, NCHW fp32) without the optional operands
    y32 = torch.zeros(B * Ho * Wo, N)
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y32, out=True, name="y32"), N, None, None, 0, None, 0,
         B, H, W, C, Ho, Wo, N, R, R, stride, pt, pl, ups, dil, 0, 1, 1, None, None, dt, tol=TOL[0] if dt == 0 else 4e-3)
    ynchw = torch.zeros(B, N, Ho, Wo)
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(ynchw, out=True, name="nchw"), 0, A(bias), None, 0, None, 0,
         B, H, W, C, Ho, Wo, N, R, R, stride, pt, pl, ups, dil, 0, 3, 1, None, None, dt, tol=TOL[0] if dt == 0 else 4e-3)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("B,Hs,C,N,pt,pl", [(8, 4, 64, 96, 2, 2), (8, 4, 64, 96, 1, 1), (8, 4, 128, 64, 2, 1), (32, 8, 128, 256, 2, 2), (128, 8, 128, 256, 1, 2),
                                            (2, 16, 64, 64, 1, 1), (128, 2, 256, 256, 2, 2)])
def test_conv_dgrad_stride2_parity_phase_form(B, Hs, C, N, pt, pl, dt):
    """Data gradient of a stride-2 3x3 conv (dilate = 1; pad 2 = forward pad 0, pad 1 = forward pad 1): when the rows of each parity phase fill
    whole 128-row tiles the generic kernel walks the output phase-major and visits only the 1 / 2 / 2 / 4 non-zero taps of a tile.  Checked
    against the emulator's zero-dilated convolution with every row-epilogue operand, pitched tensors, "+=" and the fp32 output mode; both tile
    orders (phase tiles a multiple of 8 -> XCD chunks; fewer -> natural order); the deep-ring instantiation (<= 256 blocks) and the two-buffer one."""
    H = 2 * Hs
    M = B * H * H
    assert (M // 4) % 128 == 0
    ld, yld = C + 16, N + 32
    x = r(B * Hs * Hs, ld, seed=1, dt=dt)
    w = r(N, 9 * C, seed=2, dt=dt, scale=1.0 / math.sqrt(9 * C / 4))
    bias, rowb = r(N, seed=3), r(B, N + 8, seed=4)
    res, y = r(M, yld, seed=5, dt=dt), r(M, yld, seed=6, dt=dt)
    for acc in (0, 1):
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y"), yld, A(bias), A(rowb), N + 8, A(res), yld,
             B, Hs, Hs, C, H, H, N, 3, 3, 1, pt, pl, 0, 1, acc, 0, 1, None, None, dt, tol=TOL[dt])
    # as the engine calls it: no bias, "+=" into the gradient of the block input; the caller's split-K offer is ignored
    ws, cnt = torch.zeros(-(-M // 128) * -(-N // 128) * 2 * 16384), torch.zeros(1024, dtype=torch.int32)
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y_accumulate"), yld, None, None, 0, None, 0,
         B, Hs, Hs, C, H, H, N, 3, 3, 1, pt, pl, 0, 1, 1, 0, 2, A(ws), A(cnt, out=True, name="counters"), dt, tol=TOL[dt])
    y32 = torch.zeros(M, N)
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y32, out=True, name="y32"), N, None, None, 0, None, 0,
         B, Hs, Hs, C, H, H, N, 3, 3, 1, pt, pl, 0, 1, 0, 1, 1, None, None, dt, tol=TOL[0] if dt == 0 else 4e-3)


@pytest.mark.parametrize("B,H,C,N", [(16, 32, 64, 128), (65, 16, 128, 192), (258, 8, 64, 128), (17, 32, 192, 256), (4, 64, 64, 128), (128, 8, 256, 256),
                                     (130, 8, 128, 192), (64, 8, 256, 128), (3, 24, 64, 64)])
def test_conv3x3_stationary_halo_path(B, H, C, N):
    """bf16 3x3/s1/p1 with >= 16384 output pixels takes the stationary-halo kernel: ragged image groups (B % NB != 0),
    N not a multiple of the tile, pitched operands and every epilogue fusion."""
    dt = 1
    M = B * H * H
    assert M >= 4096 or H == 24
    ld, yld = C + 16, N + 32
    x = r(M, ld, seed=1, dt=dt)
    w = r(N, 9 * C, seed=2, dt=dt, scale=1.0 / math.sqrt(9 * C))
    bias, rowb = r(N, seed=3), r(B, N + 8, seed=4)
    res, y = r(M, yld, seed=5, dt=dt), r(M, yld, seed=6, dt=dt)
    for acc in (0, 1):
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y"), yld, A(bias), A(rowb), N + 8, A(res), yld,
             B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, acc, 0, 1, None, None, dt, tol=TOL[dt])
    y2 = torch.zeros(M, N, dtype=DT[dt])
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y2, out=True, name="y_plain"), N, None, None, 0, None, 0,
         B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, None, None, dt, tol=TOL[dt])
    # "+=" without a residual (the data-gradient epilogue), bias only, time-embedding bias only
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y_accumulate"), yld, A(bias), None, 0, None, 0,
         B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 1, 0, 1, None, None, dt, tol=TOL[dt])
    both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y_rowbias"), yld, None, A(rowb), N + 8, None, 0,
         B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, None, None, dt, tol=TOL[dt])
    # the persistent kernel (conv3x3.hip) serves both patch geometries; too few pixels (3 x 24 x 24) fall through to the tile GEMMs
    assert (_hip.lib().ddpm_conv2d_variant(ld, yld, B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 1, dt, 0) in (8, 10, 13)) == (M >= 4096)


@pytest.mark.parametrize("B,Hs,C,N", [(16, 16, 128, 128), (130, 4, 128, 192), (5, 16, 256, 64), (33, 8, 64, 256)])
def test_conv3x3_upsampled_input_persistent_path(B, Hs, C, N):
    """Upsample block: nearest 2x + 3x3 / s1 / p1 in ONE launch of the persistent kernel (the gather reads stored pixel (y >> 1, x >> 1));
    both patch geometries, ragged channel tiles, pitched operands, the bias and "+=" epilogues."""
    dt, H = 1, 2 * Hs
    M = B * H * H
    ld, yld = C + 16, N + 32
    x = r(B * Hs * Hs, ld, seed=1, dt=dt)
    w = r(N, 9 * C, seed=2, dt=dt, scale=1.0 / math.sqrt(9 * C))
    bias, y = r(N, seed=3), r(M, yld, seed=6, dt=dt)
    for acc in (0, 1):
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y"), yld, A(bias), None, 0, None, 0,
             B, Hs, Hs, C, H, H, N, 3, 3, 1, 1, 1, 1, 0, acc, 0, 1, None, None, dt, tol=TOL[dt])
    assert _hip.lib().ddpm_conv2d_variant(ld, yld, B, Hs, Hs, C, H, H, N, 3, 3, 1, 1, 1, 1, 0, 0, 1, dt, 0) in (8, 10, 13)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("B,H,C,N,R", [(128, 4, 256, 256, 3), (128, 4, 512, 256, 3), (100, 4, 256, 192, 3), (16, 8, 1024, 64, 1), (128, 4, 64, 256, 3),
                                       (2, 8, 512, 512, 3), (2, 16, 512, 512, 3), (2, 4, 512, 256, 3)])
def test_conv_small_grid_split_k(B, H, C, N, R, dt):
    """Layers whose 64x64-tile grid has <= 128 blocks take two to eight K runs per tile (tiles x runs <= 256, >= 4 groups per run: the last
    three cases, the CelebA-HQ small levels at B = 2, get 8 / 4 / 8) when the caller offers the workspace (splits = 2):
    slabs + arrival counters, summed in run order by the last arriver.  Results match the emulator, the counters are back at zero,
    two launches agree bit for bit; the last case (9 K-steps of bf16: too short to split, 18 of fp32: split) covers the launcher's >= 8-groups rule both ways."""
    M = B * H * H
    ld, yld = C + 16, N + 32
    x = r(M, ld, seed=1, dt=dt)
    w = r(N, R * R * C, seed=2, dt=dt, scale=1.0 / math.sqrt(R * R * C))
    bias, rowb = r(N, seed=3), r(B, N + 8, seed=4)
    res, y = r(M, yld, seed=5, dt=dt), r(M, yld, seed=6, dt=dt)
    tiles = -(-M // 64) * -(-N // 64)
    assert tiles <= 128
    ws = torch.zeros(max(256 * 4096, tiles * 2 * 4096, -(-M // 128) * -(-N // 128) * 2 * 16384))
    cnt = torch.zeros(1024, dtype=torch.int32)
    pad = R // 2
    for acc in (0, 1):
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name="y"), yld, A(bias), A(rowb), N + 8, A(res), yld,
             B, H, H, C, H, H, N, R, R, 1, pad, pad, 0, 0, acc, 0, 2, A(ws), A(cnt, out=True, name="counters"), dt, tol=TOL[dt])
    assert _hip.lib().ddpm_conv2d_variant(ld, yld, B, H, H, C, H, H, N, R, R, 1, pad, pad, 0, 0, 0, 2, dt, 0) == 4
    xd, wd, wsd, cntd = x.cuda(), w.cuda(), ws.cuda(), cnt.cuda()
    outs = []
    for _ in range(2):
        yd = torch.zeros(M, N, dtype=DT[dt], device="cuda")
        _hip.call("ddpm_conv2d_nhwc", xd.data_ptr(), ld, wd.data_ptr(), yd.data_ptr(), N, 0, 0, 0, 0, 0,
                  B, H, H, C, H, H, N, R, R, 1, pad, pad, 0, 0, 0, 0, 2, wsd.data_ptr(), cntd.data_ptr(), dt, _hip.stream())
        outs.append(yd)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and int(cntd.abs().sum()) == 0


WG1_CASES = [(32768, 256, 256), (32768, 256, 768), (131072, 128, 256), (33635, 192, 104), (32768, 768, 256), (16384, 64, 64), (40000, 384, 128),
             (8192, 512, 256), (2048, 512, 256), (2048, 256, 256), (2050, 192, 104)]      # the 8 x 8 / 4 x 4 levels (round 5: P >= 2048 takes the slab kernel)


@pytest.mark.parametrize("P,C,N", WG1_CASES)
def test_conv1x1_wgrad_slab_kernel(P, C, N):
    """dW = dy^T x and db = colsum(dy) of a 1x1 conv by the slab kernel (csrc/wgrad1x1.hip): the slab copies, summed in order as
    ddpm_wgrad_reduce does, against float64 — full and ragged channel tiles, a pixel count that is not a multiple of the K-step,
    pitched operands."""
    splits = _hip.lib().ddpm_conv1x1_wgrad_splits(P, C, N)
    assert splits > 0
    ld_y, ld_x = N + 8, C + 16
    dy = r(P, ld_y, seed=3, dt=1, scale=0.5)
    x = r(P, ld_x, seed=4, dt=1)
    stride, bstride = N * C + 4, N + 4
    dyd, xd = dy.cuda(), x.cuda()
    slabs = torch.full((splits * stride,), float("nan"), device="cuda")
    bslabs = torch.full((splits * bstride,), float("nan"), device="cuda")
    _hip.call("ddpm_conv1x1_wgrad_nhwc", dyd.data_ptr(), ld_y, xd.data_ptr(), ld_x, slabs.data_ptr(), stride, bslabs.data_ptr(), bstride,
              P, C, N, splits, 1, _hip.stream())
    torch.cuda.synchronize()
    got = slabs.view(splits, stride)[:, :N * C].double().sum(0).view(N, C).cpu()
    gotb = bslabs.view(splits, bstride)[:, :N].double().sum(0).cpu()
    ref = dy[:, :N].double().t() @ x[:, :C].double()
    refb = dy[:, :N].double().sum(0)
    assert torch.isfinite(got).all() and torch.isfinite(gotb).all()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-3, float((got - ref).abs().max())
    assert float((gotb - refb).abs().max()) <= 2e-5 * float(refb.abs().max()) + 1e-3
    # the geometry query refuses what the kernel does not serve
    assert _hip.lib().ddpm_conv1x1_wgrad_splits(1024, C, N) == 0 and _hip.lib().ddpm_conv1x1_wgrad_splits(P, C + 4, N) == 0


PW_CASES = [(32, 32, 128, 256), (64, 32, 128, 256), (35, 31, 64, 192), (128, 16, 256, 768), (130, 16, 320, 384), (70, 31, 192, 104), (128, 16, 768, 256),
            (128, 16, 256, 256), (33, 32, 256, 128), (128, 16, 512, 256), (128, 16, 256, 512), (37, 31, 384, 128), (36, 31, 448, 96),
            (35, 31, 128, 256), (35, 31, 192, 512),        # 256-channel tiles with a ragged pixel tail
            (128, 8, 512, 256), (130, 8, 256, 512), (37, 15, 192, 384)]       # fewer pixel tiles than CUs: blocks walk (pixel, channel) tile pairs


@pytest.mark.parametrize("B,H,C,N", PW_CASES)
def test_conv1x1_streaming_path(B, H, C, N):
    """bf16 1x1 / stride 1 with >= 8192 pixels takes the persistent streaming kernel (csrc/pointwise.hip): all three tile shapes, ragged
    pixel / channel tails, one to twelve K-steps, pitched operands, and the bias / residual / accumulate epilogues."""
    dt = 1
    M = B * H * H
    assert M >= 8192 and _hip.lib().ddpm_conv2d_variant(C + 16, N + 32, B, H, H, C, H, H, N, 1, 1, 1, 0, 0, 0, 0, 0, 1, dt, 0) == 7
    ld, yld = C + 16, N + 32
    x = r(M, ld, seed=1, dt=dt)
    w = r(N, C, seed=2, dt=dt, scale=1.0 / math.sqrt(C))
    bias = r(N, seed=3)
    res, y = r(M, yld, seed=5, dt=dt), r(M, yld, seed=6, dt=dt)
    for acc, with_res, with_bias in ((0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0), (0, 0, 0)):
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y.clone(), out=True, name=f"y acc={acc} res={with_res} bias={with_bias}"), yld,
             A(bias) if with_bias else None, None, 0, A(res) if with_res else None, yld if with_res else 0,
             B, H, H, C, H, H, N, 1, 1, 1, 0, 0, 0, 0, acc, 0, 1, None, None, dt, tol=TOL[dt])


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("splits", [2, 5, 9])
def test_conv_fwd_inlaunch_splitk(dt, splits):
    """Small-M layers: K split over blocks, partial tiles reduced inside the launch by the last arriver (all epilogue
    fusions still applied once); counters must be back at zero so the next launch needs no memset."""
    B, H, C, N, R = 16, 4, 256, 256, 3
    M = B * H * H
    x, w = r(M, C, seed=1, dt=dt), r(N, 9 * C, seed=2, dt=dt, scale=0.02)
    bias, rowb, res = r(N, seed=3), r(B, N, seed=4), r(M, N, seed=5, dt=dt)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ws, cnt = torch.zeros(max(tiles * splits * 16384, 256 * 4096)), torch.zeros(tiles * 4, dtype=torch.int32)     # one counter per 64x64 tile
    for rep in range(2):
        y = r(M, N, seed=6, dt=dt)
        wsd, cntd = ws.cuda(), cnt.cuda()
        for acc in (0, 1):
            both("ddpm_conv2d_nhwc", A(x), C, A(w), A(y.clone(), out=True, name="y"), N, A(bias), A(rowb), N, A(res), N,
                 B, H, H, C, H, H, N, R, R, 1, 1, 1, 0, 0, acc, 0, splits, A(ws), A(cnt, out=True, name="cnt"), dt, tol=TOL[dt])


@pytest.mark.parametrize("dt", [0, 1])
def test_conv_out3_channels_nchw(dt):
    B, H, C, N = 2, 8, 128, 3                                   # out_conv: N = 3 rows of weights, NCHW fp32 result
    x, w = r(B * H * H, C, seed=1, dt=dt), r(N, 9 * C, seed=2, dt=dt, scale=0.03)
    y = torch.zeros(B, N, H, H)
    both("ddpm_conv2d_nhwc", A(x), C, A(w), A(y, out=True, name="y"), 0, A(r(N, seed=3)), None, 0, None, 0,
         B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 3, 1, None, None, dt, tol=TOL[0] if dt == 0 else 4e-3)


@pytest.mark.parametrize("B,H,W,C,N", [(16, 32, 32, 128, 3), (5, 64, 64, 64, 3), (17, 32, 32, 128, 8), (64, 16, 16, 256, 3)])
def test_conv_edge_few_output_channels(B, H, W, C, N):
    """out_conv on full-size images (>= 16384 pixels, bf16): the few-output-channel kernel — NCHW fp32 result, pitched input, bias or none."""
    ld = C + 16
    x, w = r(B * H * W, ld, seed=1, dt=1), r(N, 9 * C, seed=2, dt=1, scale=0.03)
    for bias in (A(r(N, seed=3)), None):
        y = torch.zeros(B, N, H, W)
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y, out=True, name="y"), 0, bias, None, 0, None, 0,
             B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 0, 0, 0, 3, 1, None, None, 1, tol=4e-3)
    assert _hip.lib().ddpm_conv2d_variant(ld, 0, B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 0, 0, 3, 1, 1, 0) == 11


@pytest.mark.parametrize("B,H,W,N", [(16, 32, 32, 128), (5, 64, 64, 64), (17, 32, 32, 96), (64, 16, 16, 128)])
def test_conv_edge_few_input_channels(B, H, W, N):
    """in_conv on full-size images (3 image channels stored as 8, bf16): the few-input-channel kernel — pitched NHWC output, bias or none."""
    C, ld, yld = 8, 8, N + 32
    x = r(B * H * W, ld, seed=1, dt=1)
    w = r(N, 9 * C, seed=2, dt=1, scale=0.2)
    for bias in (A(r(N, seed=3)), None):
        y = r(B * H * W, yld, seed=6, dt=1)
        both("ddpm_conv2d_nhwc", A(x), ld, A(w), A(y, out=True, name="y"), yld, bias, None, 0, None, 0,
             B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, None, None, 1, tol=TOL[1])
    assert _hip.lib().ddpm_conv2d_variant(ld, yld, B, H, W, C, H, W, N, 3, 3, 1, 1, 1, 0, 0, 0, 1, 1, 0) == 12


def test_gather_rows_time_table():
    """Rows of the sampler's [T][sum Cout] time-bias table by timestep; an index outside the table poisons its row."""
    T, L, B = 37, 4992, 9
    table = r(T, L, seed=1)
    idx = torch.tensor([0, 36, 5, 5, 17, 1, 36, 0, 20], dtype=torch.int64)
    both("ddpm_gather_rows_f32", A(table), A(idx), A(torch.zeros(B, L), out=True, name="rows"), B, L, T, tol=0.0)
    td, od = table.cuda(), torch.zeros(3, L, device="cuda")
    bad = torch.tensor([3, T, -1], dtype=torch.int64, device="cuda")
    _hip.call("ddpm_gather_rows_f32", td.data_ptr(), bad.data_ptr(), od.data_ptr(), 3, L, T, _hip.stream())
    torch.cuda.synchronize()
    assert torch.equal(od[0].cpu(), table[3]) and bool(torch.isnan(od[1:]).all())


WGRAD_CASES = [
    # B, H, W, C, Creal, N, Nreal, R, stride, pt, pl, ups, Ho, Wo, splits
    ("3x3", 2, 8, 8, 32, 32, 64, 64, 3, 1, 1, 1, 0, 8, 8, 1),
    ("3x3_split", 4, 16, 16, 64, 64, 128, 128, 3, 1, 1, 1, 0, 16, 16, 4),
    ("1x1", 2, 4, 4, 64, 64, 96, 96, 1, 1, 0, 0, 0, 4, 4, 1),
    ("s2", 2, 8, 8, 32, 32, 32, 32, 3, 2, 0, 0, 0, 4, 4, 2),
    ("upsample", 2, 4, 4, 32, 32, 32, 32, 3, 1, 1, 1, 1, 8, 8, 1),
    ("in_conv", 2, 8, 8, 8, 3, 32, 32, 3, 1, 1, 1, 0, 8, 8, 1),
    ("out_conv", 2, 8, 8, 32, 32, 8, 3, 3, 1, 1, 1, 0, 8, 8, 1),
]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_wgrad(case, dt):
    _, B, H, W, C, Creal, N, Nreal, R, stride, pt, pl, ups, Ho, Wo, splits = case
    if dt == 0 and (C % 4 or N % 4):
        pytest.skip("vector width")
    xld, yld = C + 8, N + 8
    x, dy = r(B * H * W, xld, seed=1, dt=dt), r(B * Ho * Wo, yld, seed=2, dt=dt)
    dw = r(Nreal, Creal, R, R, seed=3)                          # accumulates on top of existing content
    both("ddpm_conv2d_wgrad_nhwc", A(dy), yld, A(x), xld, A(dw, out=True, name="dw"), 0, B, H, W, C, Creal, Ho, Wo, N, Nreal, R, R,
         stride, pt, pl, ups, splits, dt, tol=1e-4 if dt == 0 else 3e-3)
    # slab mode: every split stores its partial into its own copy; the fixed-order reduction must give the same gradient
    K = B * Ho * Wo
    for req in (1, 3):
        eff = int(_hip.lib().ddpm_wgrad_effective_splits(K, req, dt))
        n = Nreal * Creal * R * R
        stride_f = (n + 3) // 4 * 4 + 8
        slabs = torch.full((eff * stride_f,), 7.0).cuda()             # stale content must not leak into the result
        xd, dyd = x.cuda(), dy.cuda()
        _hip.call("ddpm_conv2d_wgrad_nhwc", dyd.data_ptr(), yld, xd.data_ptr(), xld, slabs.data_ptr(), stride_f, B, H, W, C, Creal, Ho, Wo,
                  N, Nreal, R, R, stride, pt, pl, ups, eff, dt, _hip.stream())
        out = torch.zeros(n).cuda()
        table = torch.tensor([[slabs.data_ptr(), out.data_ptr(), n, eff, stride_f]], dtype=torch.int64).cuda()
        _hip.call("ddpm_wgrad_reduce", table.data_ptr(), 1, _hip.stream())
        ref = torch.zeros(n)
        Emulator().call("ddpm_conv2d_wgrad_nhwc", dy.data_ptr(), yld, x.data_ptr(), xld, ref.data_ptr(), 0, B, H, W, C, Creal, Ho, Wo,
                        N, Nreal, R, R, stride, pt, pl, ups, 1, dt, 0)
        err = float((out.cpu() - ref).abs().max())
        assert err <= (1e-4 if dt == 0 else 3e-3) * float(ref.abs().max()), (req, eff, err)
    if dt == 1:
        assert _hip.lib().ddpm_conv2d_wgrad_nhwc(dyd.data_ptr(), yld, xd.data_ptr(), xld, slabs.data_ptr(), 4, B, H, W, C, Creal, Ho, Wo,
                                                 N, Nreal, R, R, stride, pt, pl, ups, eff, dt, _hip.stream()) == 1      # stride too small


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,batch", [(16, 16, 16, 3), (256, 256, 64, 2), (300, 136, 200, 1), (128, 512, 128, 1)])
def test_gemm(dt, ta, tb, M, N, K, batch):
    v = 8 if dt else 4
    if (ta and M % v) or (tb and N % v) or K % v:
        pytest.skip("vector width")
    a = r(batch, *((K, M) if ta else (M, K)), seed=1, dt=dt)
    b = r(batch, *((K, N) if tb else (N, K)), seed=2, dt=dt)
    c = r(batch, M, N, seed=3, dt=dt)
    res, bias = r(batch, M, N, seed=4, dt=dt), r(N, seed=5)
    a_ld, b_ld = (M if ta else K), (N if tb else K)
    for acc in (0, 1):
        both("ddpm_gemm", A(a), a_ld, a[0].numel(), ta, A(b), b_ld, b[0].numel(), tb, A(c.clone(), out=True, name="c"), N, M * N,
             A(bias), A(res), N, M * N, M, N, K, batch, 0.37, acc, 0, 1, dt, tol=TOL[dt] * 4)
    c32 = r(batch, M, N, seed=6)
    both("ddpm_gemm", A(a), a_ld, a[0].numel(), ta, A(b), b_ld, b[0].numel(), tb, A(c32, out=True, name="c32"), N, M * N,
         None, None, 0, 0, M, N, K, batch, 1.0, 0, 1, 1, dt, tol=TOL[0] * 4 if dt == 0 else 4e-3)
    if K >= 128:                                                 # split-K with atomics
        c32 = r(batch, M, N, seed=7)
        both("ddpm_gemm", A(a), a_ld, a[0].numel(), ta, A(b), b_ld, b[0].numel(), tb, A(c32, out=True, name="c_atomic"), N, M * N,
             None, None, 0, 0, M, N, K, batch, 1.0, 0, 2, 2, dt, tol=TOL[0] * 4 if dt == 0 else 4e-3)


GN_CASES = [(2, 16, 32), (2, 64, 128), (3, 1024, 128), (2, 256, 384), (2, 16, 768), (1, 4096, 64), (130, 16, 256), (2, 1024, 384), (3, 256, 256), (2, 256, 512), (2, 1024, 256)]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("B,HW,C", GN_CASES)
@pytest.mark.parametrize("silu,drop", [(1, 0.0), (0, 0.0), (1, 0.1)])
def test_groupnorm_fwd_bwd(dt, B, HW, C, silu, drop):
    ld = C + 16
    x = (r(B * HW, ld, seed=1, scale=1.5) + 0.4).to(DT[dt])
    gamma, beta = 1 + 0.2 * r(C, seed=2), 0.1 * r(C, seed=3)
    nws = int(_hip.lib().ddpm_gn_workspace_floats(B, HW, C, 32, dt))
    assert nws > 0
    ws, stats = torch.zeros(nws), torch.zeros(B, 32, 2)
    y = torch.zeros(B * HW, ld, dtype=DT[dt])
    seed = 0x1234567 if drop else 0
    both("ddpm_groupnorm_silu_fwd", A(x), ld, A(y, out=True, name="y"), ld, A(gamma), A(beta), A(stats, out=True, name="stats"), A(ws),
         B, HW, C, 32, 1e-6, silu, drop, seed, None, dt, tol=TOL[dt] * (5 if dt == 0 else 1.5))
    dy = r(B * HW, ld, seed=4, dt=dt)
    # exact statistics from fp64 for the backward inputs
    xg = x.float().reshape(B, HW, ld)[:, :, :C].reshape(B, HW, 32, C // 32).permute(0, 2, 1, 3).reshape(B, 32, -1).double()
    stats = torch.stack([xg.mean(-1), 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-6)], -1).float()
    dgamma, dbeta = r(C, seed=5), r(C, seed=6)
    for acc in (0, 1):
        dx = r(B * HW, ld, seed=7, dt=dt)
        both("ddpm_groupnorm_silu_bwd", A(x), ld, A(dy), ld, A(dx, out=True, name="dx"), ld, A(gamma), A(beta), A(stats),
             A(dgamma.clone(), out=True, name="dgamma"), A(dbeta.clone(), out=True, name="dbeta"), A(ws), B, HW, C, 32, silu, drop, seed, None, acc,
             A(torch.zeros(B, C + 4), out=True, name="dx_colsum") if acc == 0 else None, C + 4,
             A(r(B * HW, ld, seed=8, dt=dt)) if acc else None, ld, dt,
             tol=2e-4 if dt == 0 else 2e-2, atol=2e-3 if dt else 1e-4)
    if drop:
        # the per-step part of the seed read from a device word (captured training step): seed + *word
        word = torch.tensor([0x5DEECE66D], dtype=torch.int64)
        y2 = torch.zeros(B * HW, ld, dtype=DT[dt])
        both("ddpm_groupnorm_silu_fwd", A(x), ld, A(y2, out=True, name="y_devseed"), ld, A(gamma), A(beta), None, A(ws),
             B, HW, C, 32, 1e-6, silu, drop, seed, A(word), dt, tol=TOL[dt] * (5 if dt == 0 else 1.5))
        dx = r(B * HW, ld, seed=7, dt=dt)
        both("ddpm_groupnorm_silu_bwd", A(x), ld, A(dy), ld, A(dx, out=True, name="dx_devseed"), ld, A(gamma), A(beta), A(stats),
             None, None, A(ws), B, HW, C, 32, silu, drop, seed, A(word), 0, None, 0, None, 0, dt, tol=2e-4 if dt == 0 else 2e-2)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("B,HW,C", [(2, 1024, 128), (2, 256, 256), (3, 64, 256), (2, 4096, 128)])
def test_groupnorm_statistics_with_mean_far_from_zero(dt, B, HW, C):
    """|mean| / std = 500 (50 +- 0.1, the case a single-pass sum / sum-of-squares loses in fp32): every path — streaming
    two-launch, staged, register-resident — must reproduce the two-pass fp64 statistics."""
    x = (r(B * HW, C, seed=1, scale=0.1) + 50.0)
    x = x + torch.arange(C).float()[None, :] * 0.01                       # channels of a group differ a little
    xq = x.to(DT[dt])
    gamma, beta = 1 + 0.2 * r(C, seed=2), 0.1 * r(C, seed=3)
    xg = xq.float().reshape(B, HW, 32, C // 32).permute(0, 2, 1, 3).reshape(B, 32, -1).double()
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    want = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-6)], -1).float()
    nws = int(_hip.lib().ddpm_gn_workspace_floats(B, HW, C, 32, dt))
    gd, bd = gamma.cuda(), beta.cuda()                        # named: a temporary's storage is recycled as soon as its pointer is taken
    for entry in ("fwd",):
        stats = torch.zeros(B, 32, 2, device="cuda")
        xd = xq.cuda()
        if entry == "fwd":
            y, ws = torch.empty_like(xd), torch.zeros(nws, device="cuda")
            _hip.call("ddpm_groupnorm_silu_fwd", xd.data_ptr(), C, y.data_ptr(), C, gd.data_ptr(), bd.data_ptr(), stats.data_ptr(),
                      ws.data_ptr(), B, HW, C, 32, 1e-6, 0, 0.0, 0, 0, dt, _hip.stream())
            ref = ((xg - mean[..., None]) / torch.sqrt(var + 1e-6)[..., None]).reshape(B, 32, HW, C // 32).permute(0, 2, 1, 3).reshape(B * HW, C)
            ref = ref * gamma.double() + beta.double()
            err = float((y.cpu().double() - ref).abs().max()) / float(ref.abs().max())
            assert err < (2e-3 if dt == 0 else 1.5e-2), (entry, err)
        got = stats.cpu()
        assert float((got[..., 0] - want[..., 0]).abs().max()) < 1e-4 * 50, entry
        assert float(((got[..., 1] - want[..., 1]) / want[..., 1]).abs().max()) < 1e-3, (entry, got[0, :3], want[0, :3])


@pytest.mark.parametrize("B,L,C", [(3, 256, 256), (2, 128, 128), (2, 384, 256)])
def test_attention_fwd_fused(B, L, C):
    ld, old = 3 * C + 16, C + 8                                # exercise pitches
    qkv = r(B * L, ld, seed=11, dt=1, scale=1.3)
    out = torch.zeros(B * L, old, dtype=torch.bfloat16)
    both("ddpm_attention_fwd", A(qkv), ld, A(out, out=True, name="o"), old, B, L, C, 1.0 / math.sqrt(C), 1, tol=TOL[1])
    # peaked rows (one dominant key): the online max / rescale path
    qkv2 = qkv.clone().float()
    qkv2[:, :C] *= 6.0
    both("ddpm_attention_fwd", A(qkv2.to(torch.bfloat16)), ld, A(out, out=True, name="o_peaked"), old, B, L, C, 1.0 / math.sqrt(C), 1, tol=TOL[1])


FLASH_CASES = [(3, 256, 256), (2, 256, 512), (2, 16, 256), (2, 64, 128), (2, 256, 128), (1, 64, 512), (2, 16, 64), (2, 128, 96), (1, 48, 32)]


@pytest.mark.parametrize("B,L,C", FLASH_CASES)
def test_attention_training_pair_keeps_no_LxL_tensor(B, L, C):
    """ddpm_attention_fwd_lse + ddpm_attention_bwd against the float64 restatement: output, log-sum-exp, D and the packed
    dq | dk | dv — at the reference's attention geometries (16x16 at C = 256 / 512, the 8x8 / 4x4 middle blocks) and ragged ones."""
    ld, old, gld = 3 * C + 16, C + 8, 3 * C + 24
    scale = 1.0 / math.sqrt(C)
    for peak in (1.0, 5.0):
        qkv = r(B * L, ld, seed=21, dt=1, scale=1.2).float()
        qkv[:, :C] *= peak
        qkv = qkv.to(torch.bfloat16)
        out = torch.zeros(B * L, old, dtype=torch.bfloat16)
        lse = torch.zeros(B * L)
        both("ddpm_attention_fwd_lse", A(qkv), ld, A(out, out=True, name="o"), old, A(lse, out=True, name="lse"), B, L, C, scale, 1, tol=TOL[1])
        both("ddpm_attention_fwd_lse", A(qkv), ld, A(out, out=True, name="o_nolse"), old, None, B, L, C, scale, 1, tol=TOL[1])
        d_o = torch.zeros(B * L, old, dtype=torch.bfloat16)
        d_o[:, :C] = r(B * L, C, seed=22, dt=1)
        dvec = torch.zeros(B * L)
        dqkv = torch.zeros(B * L, gld, dtype=torch.bfloat16)
        both("ddpm_attention_bwd", A(qkv), ld, A(out), old, A(d_o), old, A(lse), A(dvec, out=True, name="D"), A(dqkv, out=True, name="dqkv"), gld,
             B, L, C, scale, 1, tol=2e-2)
        g = dqkv.float()
        assert float(g[:, 3 * C:].abs().max()) == 0.0                        # the pitch padding is never written
        for nm, sl in (("dq", g[:, :C]), ("dk", g[:, C:2 * C]), ("dv", g[:, 2 * C:3 * C])):
            assert float(sl.abs().max()) > 0, nm


def test_attention_training_pair_rejects_what_it_cannot_serve():
    x = torch.zeros(512, 3 * 64, dtype=torch.bfloat16).cuda()
    o = torch.zeros(512, 64, dtype=torch.bfloat16).cuda()
    f = _hip.lib().ddpm_attention_fwd_lse
    assert f(x.data_ptr(), 192, o.data_ptr(), 64, 0, 1, 512, 64, 0.125, 1, _hip.stream()) == 1        # L > 256
    assert f(x.data_ptr(), 192, o.data_ptr(), 64, 0, 1, 24, 64, 0.125, 1, _hip.stream()) == 1         # L % 16
    assert f(x.data_ptr(), 192, o.data_ptr(), 64, 0, 1, 16, 48, 0.125, 1, _hip.stream()) == 1         # C % 32
    assert f(x.data_ptr(), 192, o.data_ptr(), 64, 0, 1, 16, 64, 0.125, 0, _hip.stream()) == 2         # fp32


def test_attention_fwd_rejects_unsupported_geometry():
    x = torch.zeros(16, 3 * 64, dtype=torch.bfloat16).cuda()
    o = torch.zeros(16, 64, dtype=torch.bfloat16).cuda()
    assert _hip.lib().ddpm_attention_fwd(x.data_ptr(), 192, o.data_ptr(), 64, 1, 16, 64, 0.125, 1, _hip.stream()) == 1


def test_dropout_mask_bit_exact_and_rate():
    n = 1 << 20
    m = torch.zeros(n)
    both("ddpm_dropout_mask", A(m, out=True, name="mask"), n, 0.1, 0xDEADBEEFCAFE, tol=0.0)
    d = torch.zeros(n).cuda()
    _hip.call("ddpm_dropout_mask", d.data_ptr(), n, 0.1, 0xDEADBEEFCAFE, _hip.stream())
    keep = float(d.mean())
    assert abs(keep - 0.9) < 3e-3


def test_timestep_embedding_and_layout_kernels():
    for dim in (128, 127):
        half = dim // 2
        t = torch.tensor([0, 1, 500, 999, 37])
        fr = torch.exp(-torch.arange(half, dtype=torch.float32) * (math.log(10000) / (half - 1)))
        out = torch.zeros(5, dim)
        both("ddpm_timestep_embedding", A(t), A(fr), A(out, out=True, name="temb"), 5, dim, tol=2e-5)
    for dt in (0, 1):
        x = r(2, 3, 6, 6, seed=1)
        y = torch.zeros(2 * 36, 8, dtype=DT[dt])
        both("ddpm_nchw_to_nhwc", A(x), A(y, out=True, name="nhwc"), 2, 3, 36, 8, dt, tol=TOL[dt] if dt else 0.0)
        w = r(5, 3, 3, 3, seed=2)
        wf, wd = torch.zeros(5, 9 * 8, dtype=DT[dt]), torch.zeros(3, 9 * 8, dtype=DT[dt])
        both("ddpm_pack_weight", A(w), A(wf, out=True, name="wf"), A(wd, out=True, name="wd"), 5, 3, 3, 3, 8, 8, dt, tol=TOL[dt] if dt else 0.0)


def test_diffusion_algebra_kernels():
    B, n, T = 5, 3 * 8 * 8, 1000
    x0, noise, z, out = (r(B, n, seed=s) for s in (1, 2, 3, 4))
    t = torch.tensor([0, 1, 500, 999, 250])
    tabs = [torch.rand(T, generator=torch.Generator().manual_seed(10 + i)) + 0.1 for i in range(4)]
    logvar = -torch.rand(T, generator=torch.Generator().manual_seed(20)) * 9
    xt = torch.zeros(B, n)
    both("ddpm_q_sample", A(x0), A(noise), A(t), A(tabs[0]), A(tabs[1]), A(xt, out=True, name="xt"), B, n, T, tol=1e-6)
    loss = torch.zeros(B)
    both("ddpm_mse_fwd", A(out), A(noise), A(loss, out=True, name="loss"), B, n, tol=1e-5)
    g = torch.zeros(B, n)
    both("ddpm_mse_bwd", A(out), A(noise), A(r(B, seed=5)), A(g, out=True, name="gpred"), B, n, tol=1e-6)
    for m in (1, 5, 128, 1000):               # batch mean of the per-sample losses: one block, fixed order
        tot = torch.zeros(1)
        both("ddpm_weighted_sum_f32", A(r(m, seed=6).abs()), A(torch.full((m,), 1.0 / m)), A(tot, out=True, name="mean"), m, tol=2e-6)
    for mean_type in (0, 1, 2):
        for clip in (0, 1):
            xp, px0 = torch.zeros(B, n), torch.zeros(B, n)
            both("ddpm_p_sample_step", A(x0), A(out), A(z), A(t), A(tabs[0]), A(tabs[1]), A(tabs[2]), A(tabs[3]), A(logvar),
                 A(xp, out=True, name="x_prev"), A(px0, out=True, name="pred_x0"), B, n, mean_type, clip, T, tol=2e-6)
    # NaN model output stays NaN through the clamp (torch.clamp semantics); an index outside the tables poisons its sample
    bad_out = out.clone(); bad_out[1, 5] = float("nan")
    xp = torch.zeros(B, n, device="cuda")
    dev = [v.cuda() for v in (x0, bad_out, z, t, tabs[0], tabs[1], tabs[2], tabs[3], logvar)]
    _hip.call("ddpm_p_sample_step", *[v.data_ptr() for v in dev], xp.data_ptr(), 0, B, n, 0, 1, T, _hip.stream())
    assert torch.isnan(xp[1, 5]) and int(torch.isnan(xp).sum()) == 1
    t_bad = torch.tensor([0, 1, 500, 1000, 250]).cuda()                       # 1000 is outside a table of length 1000
    dev[1], dev[3] = out.cuda(), t_bad
    _hip.call("ddpm_p_sample_step", *[v.data_ptr() for v in dev], xp.data_ptr(), 0, B, n, 0, 1, T, _hip.stream())
    assert torch.isnan(xp[3]).all() and not torch.isnan(xp[[0, 1, 2, 4]]).any()
    xq = torch.zeros(B, n, device="cuda")
    noise_d = noise.cuda()
    _hip.call("ddpm_q_sample", dev[0].data_ptr(), noise_d.data_ptr(), t_bad.data_ptr(), dev[4].data_ptr(), dev[5].data_ptr(), xq.data_ptr(), B, n, T, _hip.stream())
    assert torch.isnan(xq[3]).all() and not torch.isnan(xq[[0, 1, 2, 4]]).any()
    idx, mp, o = torch.tensor([3, 0, 49]), torch.arange(0, 1000, 20), torch.zeros(3, dtype=torch.int64)
    both("ddpm_gather_i64", A(idx), A(mp), A(o, out=True, name="gather"), 3, tol=0.0)
    tt = torch.tensor([5, 5, 5])
    both("ddpm_add_i64", A(tt, out=True, name="t"), 3, -1, tol=0.0)


@pytest.mark.parametrize("var_type", ["fixed-small", "fixed-large"])
def test_variational_bound_term_kernels(var_type):
    """`ddpm_vlb_terms` / `ddpm_vlb_terms_bwd` (loss_type = "kl": ddpm_torch/diffusion.py:203-215 with functions.py:30-65) on the REAL tables of
    the linear schedule, a batch that mixes t = 0 (discretized-Gaussian NLL, incl. both open tails |x_0| > 0.999) with t > 0 (KL between the
    posterior and the model), all three mean parameterisations, clipped and unclipped estimates — against the emulator's torch restatement
    (itself checked against the oracle on CPU, the oracle against the live reference)."""
    from oracle import diffusion_ref as D
    B, n, T = 6, 3 * 8 * 8, 1000
    tb = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, T), var_type)
    tabs = [tb[k].float() for k in ("sqrt_recip_alphas_bar", "sqrt_recip_m1_alphas_bar", "posterior_mean_coef1", "posterior_mean_coef2",
                                   "posterior_logvar_clipped", "fixed_model_logvar")]
    x0 = (r(B, n, seed=1) * 0.5).clamp(-1, 1)
    x0[0, :4] = torch.tensor([1.0, -1.0, 0.9995, -0.9995])
    t = torch.tensor([0, 0, 1, 500, 999, 250])
    noise = r(B, n, seed=2)
    xt = tb["sqrt_alphas_bar"].float()[t][:, None] * x0 + tb["sqrt_one_minus_alphas_bar"].float()[t][:, None] * noise
    gl = r(B, seed=5)
    for mean_type in (0, 1, 2):
        # an output in the neighbourhood of a sensible prediction, so that the t = 0 rows are not all in the floored region
        out = (noise if mean_type == 0 else x0 if mean_type == 1 else tabs[2][t][:, None] * x0 + tabs[3][t][:, None] * xt) + 0.05 * r(B, n, seed=7)
        for clip in (0, 1):
            loss, px0 = torch.zeros(B), torch.zeros(B, n)
            both("ddpm_vlb_terms", A(x0), A(xt), A(out), A(t), *[A(v) for v in tabs], A(loss, out=True, name="bpd"), A(px0, out=True, name="pred_x0"),
                 B, n, mean_type, clip, T, tol=5e-5)
        g = torch.zeros(B, n)
        both("ddpm_vlb_terms_bwd", A(x0), A(xt), A(out), A(t), *[A(v) for v in tabs], A(gl), A(g, out=True, name="gout"), B, n, mean_type, T,
             tol=2e-4)
    # an index outside the tables poisons its row only
    dev = [v.cuda() for v in (x0, xt, out, torch.tensor([0, 1, 1000, 5, 7, 9]))] + [v.cuda() for v in tabs]
    loss = torch.zeros(B, device="cuda")
    _hip.call("ddpm_vlb_terms", *[v.data_ptr() for v in dev], loss.data_ptr(), 0, B, n, 0, 0, T, _hip.stream())
    assert torch.isnan(loss[2]) and not torch.isnan(loss[[0, 1, 3, 4, 5]]).any()


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("B,H,W,C", [(3, 4, 4, 64), (2, 8, 6, 136), (130, 2, 2, 256)])
def test_resample2x_pool_and_replicate(B, H, W, C, dt):
    """resample_with_conv=False (ddpm_torch/models/unet.py:169 AvgPool2d(2), :196 bare nearest Upsample) and their autograd: one kernel, both
    directions, pitched tensors, "=" and "+=", the 1/4 of the average as `scale`."""
    xl, yl = C + 16, C + 8
    big, small = r(B * 4 * H * W, xl, seed=1, dt=dt), r(B * H * W, yl, seed=2, dt=dt)
    for acc in (0, 1):
        both("ddpm_resample2x_nhwc", A(big), xl, A(small.clone(), out=True, name="pooled"), yl, B, H, W, C, 0, 0.25, acc, dt, tol=TOL[dt])
        both("ddpm_resample2x_nhwc", A(small), yl, A(big.clone(), out=True, name="replicated"), xl, B, H, W, C, 1, 1.0, acc, dt, tol=TOL[dt])
    both("ddpm_resample2x_nhwc", A(small), yl, A(big.clone(), out=True, name="pool_bwd"), xl, B, H, W, C, 1, 0.25, 1, dt, tol=TOL[dt])
    assert _hip.lib().ddpm_resample2x_nhwc(0, xl, 1, yl, B, H, W, C, 0, ctypes.c_float(1.0), 0, dt, 0) == 5
    assert _hip.lib().ddpm_resample2x_nhwc(1, xl, 1, yl, B, H, W, C + 1, 0, ctypes.c_float(1.0), 0, dt, 0) == 1


@pytest.mark.parametrize("dt", [0, 1])
def test_reductions_and_fanin(dt):
    B, HW, C, ld = 3, 64, 256, 272
    dy = r(B * HW, ld, seed=1, dt=dt)
    ps, tot = r(B, C + 4, seed=12), r(C, seed=2)
    both("ddpm_colsum", A(dy), ld, A(ps, out=True, name="per_sample"), C + 4, A(tot, out=True, name="total"), B, HW, C, dt, tol=2e-5 if dt == 0 else 1e-4)
    # wide tensors in one launch (channel groups of 256 vectors on blockIdx.z, a ragged last group): the [B][sum Cout] time-bias gradient
    Cw = 4992 if dt == 0 else 2 * 2048 + 136
    dyw, psw, totw = r(2 * 128, Cw + 8, seed=7, dt=dt), r(2, Cw + 4, seed=8), r(Cw, seed=9)
    both("ddpm_colsum", A(dyw), Cw + 8, A(psw, out=True, name="per_sample_wide"), Cw + 4, A(totw, out=True, name="total_wide"), 2, 128, Cw, dt,
         tol=2e-5 if dt == 0 else 1e-4)
    up = r(B * 4 * 16, 64, seed=3, dt=dt)
    for acc in (0, 1):
        dx = r(B * 16, 80, seed=4, dt=dt)
        both("ddpm_upsample2x_bwd", A(up), A(dx, out=True, name="dx"), 80, B, 4, 4, 64, acc, dt, tol=TOL[dt])
        y = r(40, 80, seed=5, dt=dt)
        both("ddpm_add_rows", A(r(40, 72, seed=6, dt=dt)), 72, A(y, out=True, name="y"), 80, 40, 64, acc, dt, tol=TOL[dt])
    for L in (16, 64, 256):
        s = r(6 * L, L, seed=7, scale=2.0)
        p = torch.zeros(6 * L, L, dtype=DT[dt])
        both("ddpm_softmax_fwd", A(s), A(p, out=True, name="p"), 6 * L, L, dt, tol=TOL[dt])
        pp = torch.softmax(s, -1).to(DT[dt])
        ds = torch.zeros(6 * L, L, dtype=DT[dt])
        both("ddpm_softmax_bwd", A(pp), A(r(6 * L, L, seed=8)), A(ds, out=True, name="ds"), 6 * L, L, dt, tol=TOL[dt] * 2)
    x = r(1000, seed=9)
    y = torch.zeros(1000)
    both("ddpm_silu_fwd", A(x), A(y, out=True, name="silu"), 1000, tol=1e-6)
    for acc in (0, 1):
        dx = r(1000, seed=10)
        both("ddpm_silu_bwd", A(x), A(r(1000, seed=11)), A(dx, out=True, name="dsilu"), 1000, acc, tol=2e-6)


def test_fused_adam_ema_matches_torch_adam():
    n = 10007
    p0, g = r(n, seed=1), r(n, seed=2, scale=3.0)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    p, m, v, sh = p0.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), p0.cuda().clone()
    shadow_ref = p0.clone()
    tot, ws = torch.zeros(1).cuda(), torch.zeros(1024).cuda()
    for step in range(1, 4):
        gs = g * step
        ref.grad = gs.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        d = min(0.9999, step / (9 + step))
        shadow_ref += (1 - d) * (ref.detach() - shadow_ref)
        gd = gs.cuda()
        tot.zero_()
        _hip.call("ddpm_sumsq_accumulate", gd.data_ptr(), n, tot.data_ptr(), ws.data_ptr(), _hip.stream())
        _hip.call("ddpm_adam_ema_step", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, tot.data_ptr(), 1.0,
                  2e-4, 0.9, 0.999, 1e-8, 1 - 0.9 ** step, 1 - 0.999 ** step, 1 - d, _hip.stream())
        assert abs(float(tot.sqrt()) - float(gs.norm())) < 1e-3 * float(gs.norm())
    assert float((p.cpu() - ref.detach()).abs().max()) < 2e-6
    assert float((sh.cpu() - shadow_ref).abs().max()) < 2e-6


def test_multi_tensor_update_with_scalars_from_device_memory():
    """ddpm_mt_grad_sumsq + ddpm_mt_adam_ema over a pointer table, scalars by value and from the device `hyper` words (the
    captured training step), against torch.optim.Adam + clip_grad_norm_ + the EMA recurrence."""
    sizes = [4096, 37, 1000, 128 * 128 * 9, 3]
    ps = [r(n, seed=10 + i) for i, n in enumerate(sizes)]
    refs = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(refs, lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
    shadow_ref = [p.clone() for p in ps]
    for hyper_mode in (False, True):
        P = [p.clone().cuda() for p in ps]
        M, V, S = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P], [p.clone() for p in P]
        G = [torch.empty_like(p) for p in P]
        table = torch.tensor([[p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), s_.data_ptr(), p.numel()] for p, g, m, v, s_ in zip(P, G, M, V, S)],
                             dtype=torch.int64).cuda()
        total = torch.full((_hip.lib().ddpm_mt_sumsq_slots(len(sizes)),), 7.0, device="cuda")     # dirty on entry: the bank is OVERWRITTEN
        hyper = torch.zeros(4, device="cuda")
        if not hyper_mode:
            refs = [p.clone().requires_grad_(True) for p in ps]
            opt = torch.optim.Adam(refs, lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
            shadow_ref = [p.clone() for p in ps]
        else:
            refs = [p.clone().requires_grad_(True) for p in ps]
            opt = torch.optim.Adam(refs, lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
            shadow_ref = [p.clone() for p in ps]
        for step in range(1, 4):
            gs = [r(n, seed=100 * step + i, scale=0.5) for i, n in enumerate(sizes)]
            for q, g_ in zip(refs, gs):
                q.grad = g_.clone()
            norm = torch.nn.utils.clip_grad_norm_(refs, 1.0)
            opt.step()
            d = min(0.9999, step / (9 + step))
            for sr, q in zip(shadow_ref, refs):
                sr += (1 - d) * (q.detach() - sr)
            for g_dev, g_ in zip(G, gs):
                g_dev.copy_(g_)
            _hip.call("ddpm_mt_grad_sumsq", table.data_ptr(), len(P), total.data_ptr(), total.numel(), _hip.stream())
            first = total[:64].clone()
            assert abs(float(first.sum().sqrt()) - float(norm)) < 1e-4 * float(norm) and float(first[1:].abs().max()) == 0.0
            _hip.call("ddpm_mt_grad_sumsq", table.data_ptr(), len(P), total.data_ptr(), total.numel(), _hip.stream())
            assert torch.equal(total[:64], first)                              # fixed-order sum: the same bits every time
            sc = (3e-3, 1 - 0.9 ** step, 1 - 0.999 ** step, 1 - d)
            if hyper_mode:
                hyper.copy_(torch.tensor(sc))
                _hip.call("ddpm_mt_adam_ema", table.data_ptr(), len(P), total.data_ptr(), 1.0, 0.0, 0.9, 0.999, 1e-8, 1.0, 1.0, 0.0, hyper.data_ptr(), _hip.stream())
            else:
                _hip.call("ddpm_mt_adam_ema", table.data_ptr(), len(P), total.data_ptr(), 1.0, *sc[:1], 0.9, 0.999, 1e-8, *sc[1:], 0, _hip.stream())
        for p, q, s_, sr in zip(P, refs, S, shadow_ref):
            assert float((p.cpu() - q.detach()).abs().max()) < 3e-6, hyper_mode
            assert float((s_.cpu() - sr).abs().max()) < 3e-6, hyper_mode


def test_multi_tensor_gather():
    a = [r(n, seed=i) for i, n in enumerate((512 * 128, 128, 7, 4096))]
    b = [r(n, seed=10 + i) for i, n in enumerate((512 * 128, 128, 7, 4096))]
    A_, B_ = [t.cuda() for t in a], [t.cuda() for t in b]
    dst = torch.zeros(sum(t.numel() for t in a) + 8, device="cuda")
    rows, off = [], 0
    for i, (x, y) in enumerate(zip(A_, B_)):
        rows.append([x.data_ptr(), y.data_ptr() if i % 2 else 0, dst.data_ptr() + 4 * off, x.numel()])
        off += x.numel()
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    _hip.call("ddpm_mt_gather_f32", table.data_ptr(), len(rows), _hip.stream())
    want = torch.cat([x + y if i % 2 else x for i, (x, y) in enumerate(zip(a, b))])
    assert torch.equal(dst[:off].cpu(), want) and float(dst[off:].abs().max()) == 0


WG3_CASES = [
    # B, H, W, C, N, Nreal, splits
    ("32x32", 3, 32, 32, 64, 128, 128, 0),
    ("16x16_wide", 2, 16, 16, 128, 256, 256, 0),
    ("16x48_ragged_n", 2, 16, 48, 32, 72, 72, 3),
    ("8x8_odd_batch", 5, 8, 8, 64, 64, 64, 0),
    ("4x4_ragged", 9, 4, 4, 32, 96, 96, 2),
    ("64x64_one_slice", 1, 64, 64, 32, 64, 60, 1),
    ("full_batch_cifar_level0", 128, 32, 32, 128, 128, 128, 0),
]


@pytest.mark.parametrize("up", [0, 1], ids=["same_size", "upsampled_input"])
@pytest.mark.parametrize("case", WG3_CASES, ids=[c[0] for c in WG3_CASES])
def test_conv3x3_wgrad_patch_kernel(case, up):
    """Patch-stationary 3x3 weight gradient (+ folded bias gradient) against the emulator: atomics mode on top of existing
    content, slab mode over stale copies followed by the fixed-order reduction; pitched operands, ragged image groups,
    out-channel tiles that run past N.  up = 1: the Upsample block's conv (x stored at half of dy's image size, nearest-2x gather in the kernel)."""
    _, B, H, W, C, N, Nreal, splits = case
    dt = 1
    xld, yld = C + 8, N + 16
    fn = "ddpm_conv3x3_wgrad_up_nhwc" if up else "ddpm_conv3x3_wgrad_nhwc"
    x, dy = r(B * (H >> up) * (W >> up), xld, seed=1, dt=dt), r(B * H * W, yld, seed=2, dt=dt)
    copies = int(_hip.lib().ddpm_conv3x3_wgrad_splits(B, H, W, C, N, splits))
    assert copies >= 1
    n = Nreal * 9 * C
    ref_w, ref_b = torch.zeros(n), torch.zeros(Nreal)
    Emulator(_hip.lib()).call(fn, dy.data_ptr(), yld, x.data_ptr(), xld, ref_w.data_ptr(), 0, ref_b.data_ptr(), 0,
                              B, H, W, C, N, Nreal, splits, dt, 0)
    xd, dyd = x.cuda(), dy.cuda()
    tol = 4e-3
    # atomics on top of existing content
    w0, b0 = r(n, seed=3), r(Nreal, seed=4)
    wd, bd = w0.cuda(), b0.cuda()
    _hip.call(fn, dyd.data_ptr(), yld, xd.data_ptr(), xld, wd.data_ptr(), 0, bd.data_ptr(), 0, B, H, W, C, N, Nreal, splits, dt, _hip.stream())
    assert float((wd.cpu() - w0 - ref_w).abs().max()) <= tol * float(ref_w.abs().max()), "atomic dw"
    assert float((bd.cpu() - b0 - ref_b).abs().max()) <= tol * float(ref_b.abs().max()), "atomic dbias"
    # slab copies + fixed-order reduction, twice: bit-identical
    stride, bstride = (n + 3) // 4 * 4 + 8, (Nreal + 3) // 4 * 4 + 4
    outs = []
    for rep in range(2):
        slabs = torch.full((copies * (stride + bstride),), 7.0 + rep).cuda()          # stale content must not leak
        bptr = slabs.data_ptr() + 4 * copies * stride
        _hip.call(fn, dyd.data_ptr(), yld, xd.data_ptr(), xld, slabs.data_ptr(), stride, bptr, bstride,
                  B, H, W, C, N, Nreal, splits, dt, _hip.stream())
        out_w, out_b = torch.zeros(n).cuda(), torch.zeros(Nreal).cuda()
        table = torch.tensor([[slabs.data_ptr(), out_w.data_ptr(), n, copies, stride], [bptr, out_b.data_ptr(), Nreal, copies, bstride]], dtype=torch.int64).cuda()
        _hip.call("ddpm_wgrad_reduce", table.data_ptr(), 2, _hip.stream())
        outs.append((out_w.cpu(), out_b.cpu()))
    assert float((outs[0][0] - ref_w).abs().max()) <= tol * float(ref_w.abs().max()), "slab dw"
    assert float((outs[0][1] - ref_b).abs().max()) <= tol * float(ref_b.abs().max()), "slab dbias"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])       # deterministic
    # without a bias pointer nothing else changes
    w2 = torch.zeros(n).cuda()
    _hip.call(fn, dyd.data_ptr(), yld, xd.data_ptr(), xld, w2.data_ptr(), 0, 0, 0, B, H, W, C, N, Nreal, splits, dt, _hip.stream())
    assert float((w2.cpu() - ref_w).abs().max()) <= tol * float(ref_w.abs().max())


def test_conv3x3_wgrad_patch_kernel_rejects_unsupported_geometry():
    lib = _hip.lib()
    assert lib.ddpm_conv3x3_wgrad_splits(2, 32, 32, 8, 128, 0) == 0          # C not a multiple of 32 (in_conv)
    assert lib.ddpm_conv3x3_wgrad_splits(2, 12, 12, 64, 64, 0) == 0          # neither 4x4 / 8x8 nor multiples of 16
    assert lib.ddpm_conv3x3_wgrad_splits(2, 32, 32, 64, 3, 0) == 0           # N not a multiple of 8 (out_conv, unpadded)
    t = torch.zeros(64, device="cuda")
    assert lib.ddpm_conv3x3_wgrad_nhwc(t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 0, 0, 0, 2, 12, 12, 64, 64, 64, 0, 1, _hip.stream()) == 1
    assert lib.ddpm_conv3x3_wgrad_nhwc(t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 0, 0, 0, 2, 16, 16, 64, 64, 64, 0, 0, _hip.stream()) == 2     # fp32


def test_wgrad_unpack_and_fused_norm():
    """ddpm_wgrad_unpack / ddpm_wgrad_unpack_sumsq: packed [N][R*S][C] -> [N][C][R*S] (x scale) for several tensors in one launch, plain
    segment copies (R*S = 1), and — fused form — the squared norm of everything written, against float64."""
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 48, 9), (7, 24, 9), (1, 1000, 1), (33, 5, 1), (16, 16, 16), (37, 128, 9), (256, 64, 9), (5, 192, 9)]     # the last three: the wave-slab path (C % 64 == 0)
    src_off, dst_off, rows = 0, 0, []
    for N, C, RS in shapes:
        rows.append([src_off, dst_off, N, C, RS])
        src_off += (N * C * RS + 3) // 4 * 4 + 4
        dst_off += (N * C * RS + 3) // 4 * 4
    gpack = torch.randn(src_off, generator=g)
    descs = torch.tensor(rows, dtype=torch.int64)
    for fused in (0, 1):
        gflat = torch.full((dst_off,), 3.0)
        slots = _hip.lib().ddpm_mt_sumsq_slots(len(rows))
        total = torch.zeros(slots)
        if fused:
            both("ddpm_wgrad_unpack_sumsq", A(gpack), A(gflat, out=True, name="gflat"), A(descs), len(rows), 0.5, A(total), slots, tol=0.0)
            dp, df, dd, dtot = gpack.cuda(), torch.zeros(dst_off).cuda(), descs.cuda(), torch.full((slots,), 7.0).cuda()     # dirty on entry
            ref = sum(float((0.5 * gpack[s:s + N * C * RS].double()).pow(2).sum()) for s, _, N, C, RS in rows)
            seen = []
            for _ in range(3):
                _hip.call("ddpm_wgrad_unpack_sumsq", dp.data_ptr(), df.data_ptr(), dd.data_ptr(), len(rows), 0.5, dtot.data_ptr(), slots, _hip.stream())
                seen.append(dtot[:64].clone())
                assert abs(float(dtot[0].double()) - ref) <= 1e-5 * ref and float(dtot[1:64].abs().max()) == 0.0
            assert torch.equal(seen[0], seen[1]) and torch.equal(seen[0], seen[2])       # fixed-order sum: the same bits every launch
            # a buffer sized for the pre-round-5 contract (a 64-float bank) is refused, not written past
            lib = _hip.lib()
            assert lib.ddpm_wgrad_unpack_sumsq(dp.data_ptr(), df.data_ptr(), dd.data_ptr(), len(rows), 0.5, dtot.data_ptr(), 64, _hip.stream()) == 1
            assert lib.ddpm_mt_grad_sumsq(dd.data_ptr(), len(rows), dtot.data_ptr(), slots - 1, _hip.stream()) == 1
        else:
            both("ddpm_wgrad_unpack", A(gpack), A(gflat, out=True, name="gflat"), A(descs), len(rows), 0.5, tol=0.0)


@pytest.mark.parametrize("dt", [0, 1])
def test_pack_weight_multi(dt):
    """All derived weight layouts in one launch: forward pack [N][tap][Cp], dgrad pack [C][flipped tap][Np], the 4x4 / stride-2 effective
    dgrad kernel of the Upsample convs, channel padding, ragged 32-tiles — bit-exact against the numpy restatement (the 4x4 kernel, a sum
    of up to four fp32 taps, to fp32 / bf16 rounding)."""
    vec = 4 if dt == 0 else 8
    pad = lambda n: -(-n // vec) * vec
    cases = [(64, 64, 3, 0), (40, 24, 3, 0), (3, 128, 3, 0), (128, 3, 3, 0), (96, 72, 1, 0), (64, 64, 3, 0x100), (40, 72, 3, 0x100)]
    ws, wfs, wds, rows, keep = [], [], [], [], []
    for i, (N, C, R, flag) in enumerate(cases):
        w = r(N, C, R, R, seed=10 + i)
        Cp, Np = pad(C), pad(N)
        wf = torch.zeros(N * R * R * Cp, dtype=DT[dt])
        wd = torch.zeros(C * (16 if flag else R * R) * Np, dtype=DT[dt])
        ws.append(w); wfs.append(wf); wds.append(wd)
    host = [(w.data_ptr(), wf.data_ptr(), wd.data_ptr()) for w, wf, wd in zip(ws, wfs, wds)]
    dev = [(w.cuda(), wf.cuda(), wd.cuda()) for w, wf, wd in zip(ws, wfs, wds)]
    mk = lambda ptrs: torch.tensor([[p[0], p[1], p[2], N, C, R | flag, pad(C), pad(N)] for p, (N, C, R, flag) in zip(ptrs, cases)], dtype=torch.int64)
    htab = mk(host)                                                   # (kept alive: the emulator reads it through its address)
    Emulator().call("ddpm_pack_weight_multi", htab.data_ptr(), len(cases), dt, 0)
    dtab = mk([(a.data_ptr(), b.data_ptr(), c.data_ptr()) for a, b, c in dev]).cuda()
    _hip.call("ddpm_pack_weight_multi", dtab.data_ptr(), len(cases), dt, _hip.stream())
    torch.cuda.synchronize()
    for (N, C, R, flag), wf, wd, (_, dwf, dwd) in zip(cases, wfs, wds, dev):
        assert torch.equal(dwf.cpu(), wf), (N, C, R, flag, "wf")
        if flag:
            assert float((dwd.cpu().float() - wd.float()).abs().max()) <= (1e-6 if dt == 0 else 2e-2) * float(wd.float().abs().max()), (N, C, "wd 4x4")
        else:
            assert torch.equal(dwd.cpu(), wd), (N, C, R, flag, "wd")


@pytest.mark.parametrize("B,H,C,N,res", [(16, 32, 64, 128, 0), (128, 32, 128, 128, 0), (128, 16, 256, 256, 0), (72, 16, 192, 192, 0), (128, 16, 256, 256, 1)])
def test_conv3x3_wave_specialised_kernel_is_bit_stable_under_contention(B, H, C, N, res):
    """conv3x3_pc_kernel synchronises its loader and consumer waves through counters in LDS, no barrier: its result must not depend on
    how the waves of a block happen to progress.  Run alone -> reference bits; then 40 launches next to an MFMA-only kernel that holds a
    wave on every SIMD of every CU (ddpm_mfma_probe on a second stream: the block's waves then advance unevenly) -> the same bits every
    time.  (A first version summed the four loaders' `ready` signals in one counter: a loader running ahead stood in for one that was
    behind, and a tile's quarter could be multiplied before it had landed — rare, and only under contention.)"""
    dt = 1
    M = B * H * H
    x = r(M, C, seed=11, dt=dt).to(DEV)
    w = r(N, 9 * C, seed=12, dt=dt, scale=1.0 / math.sqrt(9 * C)).to(DEV)
    bias = r(N, seed=13).to(DEV)
    resid = r(M, N, seed=14, dt=dt).to(DEV)
    lib = _hip.lib()
    # the wave-specialised kernel takes the calls without a residual / "+=" epilogue; those stay on the round-3 kernel (same bits)
    assert lib.ddpm_conv2d_variant(C, N, B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 1, dt, 0) == 13
    assert lib.ddpm_conv2d_variant(C, N, B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 1, dt, 1) == 8
    assert lib.ddpm_conv2d_variant(C, N, B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 1, dt, 2) == 8

    def conv(y, stream):
        _hip.call("ddpm_conv2d_nhwc", x.data_ptr(), C, w.data_ptr(), y.data_ptr(), N, bias.data_ptr(), 0, 0, resid.data_ptr() if res else 0, N if res else 0,
                  B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, dt, stream)
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    conv(ref, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sink = torch.zeros(256, device=DEV)
    outs = [torch.empty_like(ref) for _ in range(40)]
    for it in range(2):
        _hip.call("ddpm_mfma_probe", sink.data_ptr(), 60000, 0, s1.cuda_stream)        # ~10 ms of matrix-pipe contention on every CU
        for y in outs[it * 20:(it + 1) * 20]:
            conv(y, s2.cuda_stream)
        torch.cuda.synchronize()
    bad = [i for i, y in enumerate(outs) if not torch.equal(y, ref)]
    assert not bad, f"launches {bad} differ from the uncontended result"


@pytest.mark.parametrize("B,H,C,N,up", [(16, 32, 128, 128, 0), (24, 16, 256, 256, 0), (8, 32, 256, 128, 1)])
def test_wgrad3x3_wave_specialised_kernel_is_bit_stable_under_contention(B, H, C, N, up):
    """wgrad3x3_ws_kernel (round 6) synchronises four loader and four consumer waves through single-writer counters in LDS — no stage
    barrier — and keeps only two stages in flight: a consumer that read a slot before every loader's share had landed, or a loader that
    refilled it before every consumer was through, would show up as different bits.  Slab copies alone -> reference; then 30 launches next
    to an MFMA-only kernel holding a wave on every SIMD (the block's waves advance unevenly) -> the same bits every time; no wait expired."""
    dt = 1
    lib = _hip.lib()
    assert lib.ddpm_conv3x3_wgrad_variant(B, H, H, C, N) == 14
    x = r(B * (H >> up) * (H >> up), C, seed=21, dt=dt).to(DEV)
    dy = r(B * H * H, N, seed=22, dt=dt).to(DEV)
    copies = int(lib.ddpm_conv3x3_wgrad_splits(B, H, H, C, N, 0))
    n = N * 9 * C
    fn = "ddpm_conv3x3_wgrad_up_nhwc" if up else "ddpm_conv3x3_wgrad_nhwc"

    def run(slab, stream):
        _hip.call(fn, dy.data_ptr(), N, x.data_ptr(), C, slab.data_ptr(), n, slab.data_ptr() + 4 * copies * n, N, B, H, H, C, N, N, 0, dt, stream)
    ref = torch.full((copies * (n + N),), 3.0, device=DEV)
    run(ref, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sink = torch.zeros(256, device=DEV)
    outs = [torch.full_like(ref, 5.0) for _ in range(30)]
    for it in range(2):
        _hip.call("ddpm_mfma_probe", sink.data_ptr(), 60000, 0, s1.cuda_stream)
        for y in outs[it * 15:(it + 1) * 15]:
            run(y, s2.cuda_stream)
        torch.cuda.synchronize()
    bad = [i for i, y in enumerate(outs) if not torch.equal(y, ref)]
    assert not bad, f"launches {bad} differ from the uncontended result"
    fault = (ctypes.c_uint * 4)()
    assert lib.ddpm_wgrad3x3_ws_last_fault(fault) == 0 and fault[3] == 0


@pytest.mark.parametrize("M,N,K", [(1024, 512, 128), (256, 512, 128), (512, 128, 128), (2048, 512, 2), (130, 70, 37), (100, 72, 96), (36, 260, 128),
                                   (256, 128, 200), (64, 64, 129)])
def test_atb_f32_short_reduction_product(M, N, K):
    """C = A^T B in fp32 for short reductions (csrc/elementwise.hip: the time-embedding path's weight gradients, K = the batch) against float64,
    with pitched operands and ragged tile edges; two launches are bit-identical (plain stores, fixed order).  K <= 128 with 4-element-aligned
    shapes takes `atb_f32_short_kernel` (32 x 64 tiles, the whole reduction fetched at once; ragged tiles: 100 x 72, 36 x 260), everything
    else — odd extents, K > 128 — `atb_f32_kernel`."""
    lda, ldb, ldc = M + 12, N + 4, N + 8
    g = torch.Generator().manual_seed(11)
    a = torch.randn(K, lda, generator=g)
    b = torch.randn(K, ldb, generator=g)
    ad, bd = a.cuda(), b.cuda()
    outs = []
    for _ in range(2):
        c = torch.full((M, ldc), float("nan"), device="cuda")
        _hip.call("ddpm_atb_f32", ad.data_ptr(), lda, bd.data_ptr(), ldb, c.data_ptr(), ldc, M, N, K, _hip.stream())
        torch.cuda.synchronize()
        outs.append(c.cpu())
    got = outs[0][:, :N]
    ref = a[:, :M].double().t() @ b[:, :N].double()
    assert torch.equal(outs[0][:, :N], outs[1][:, :N]) and torch.isnan(outs[0][:, N:]).all()      # nothing written beyond column N
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * max(K, 8) ** 0.5
    assert _hip.lib().ddpm_atb_f32(0, lda, bd.data_ptr(), ldb, 1, ldc, M, N, K, 0) == 5 and _hip.lib().ddpm_atb_f32(ad.data_ptr(), M - 1, bd.data_ptr(), ldb, 1, ldc, M, N, K, 0) == 1
