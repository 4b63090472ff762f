"""Thin Python wrappers over the C-ABI (one function per entry point family).

Activations are NHWC; a ``View`` is (base tensor kept alive, device pointer, pixel pitch, dims) so that
channel slices of a shared buffer (the zero-copy decoder concat, q/k/v inside project_in's output) are
first-class operands.  Nothing here allocates or synchronises except where noted.
"""
import os

import torch

from . import _hip

GN_GROUPS = 32
GN_EPS = 1e-6

# Optional per-launch timing (bench.py's roofline leg): when a list, every MFMA launch appends
# (kind, algorithmic flops, start event, end event, shape, kernel id) recorded on the launch stream; kinds starting with "hbm_" are the
# HBM-bound GroupNorm launches, whose second field is algorithmic BYTES.
PROFILE = None


def _timed(kind, flops, fn, shape="", variant=None):
    """variant: callable returning the id of the kernel the library dispatches this call to (a stateless query)."""
    if PROFILE is None:
        fn()
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the events go on the stream the kernel goes to: a routed side stream is not torch's current stream
    st = torch.cuda.ExternalStream(_hip._routed) if _hip._routed is not None else torch.cuda.current_stream()
    a.record(st)
    fn()
    b.record(st)
    PROFILE.append((kind, flops, a, b, shape, variant() if variant is not None else 0))


class View:
    """NHWC activation view: element (b, y, x, c) lives at ptr + ((b*H + y)*W + x)*ld + c (in elements)."""
    __slots__ = ("base", "ptr", "ld", "B", "H", "W", "C", "dtype", "grad", "ginit")

    def __init__(self, base, B, H, W, C, ld=None, offset=0):
        self.base = base
        self.ptr = base.data_ptr() + offset * base.element_size()
        self.ld = C if ld is None else ld
        self.B, self.H, self.W, self.C = B, H, W, C
        self.dtype = _hip.dt(base)
        self.grad = None        # View holding d(loss)/d(this) during backward
        self.ginit = False      # has .grad been written yet (else the first writer stores, later ones accumulate)

    @staticmethod
    def new(B, H, W, C, dtype, device):
        return View(_hip.retain(torch.empty((B, H, W, C), dtype=dtype, device=device)), B, H, W, C)

    def chan_slice(self, c0, c1):
        v = View.__new__(View)
        v.base = self.base
        v.ptr = self.ptr + c0 * self.base.element_size()
        v.ld, v.B, v.H, v.W, v.C, v.dtype = self.ld, self.B, self.H, self.W, c1 - c0, self.dtype
        v.grad, v.ginit = None, False
        return v

    @property
    def rows(self):
        return self.B * self.H * self.W

    def to_nchw(self):
        """Debug/test helper: materialise as a contiguous NCHW fp32 tensor (uses torch indexing, not a kernel)."""
        es = self.base.element_size()
        off = (self.ptr - self.base.data_ptr()) // es
        flat = self.base.reshape(-1)
        idx = torch.as_strided(flat, (self.B, self.H, self.W, self.C), (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1), off)
        return idx.permute(0, 3, 1, 2).float().contiguous()


def conv2d(x, w_ptr, y_ptr, y_ld, N, R, S, Ho, Wo, stride=1, pad_t=0, pad_l=0, upsample=0, dilate=0,
           bias=0, rowbias=0, rowbias_ld=0, res_ptr=0, res_ld=0, accumulate=0, out_mode=0, splitk=None):
    """x: View (its H, W are the STORED input dims).  splitk: SplitK workspace (small-M layers get an in-launch split-K)."""
    splits, ws, cnt = splitk.plan(x.B * Ho * Wo, N, R * S * x.C, x.dtype) if splitk is not None else (1, 0, 0)
    _timed("gemm_nn", 2.0 * x.B * Ho * Wo * N * R * S * x.C, lambda: _hip.call(
        "ddpm_conv2d_nhwc", x.ptr, x.ld, w_ptr, y_ptr, y_ld, bias, rowbias, rowbias_ld, res_ptr, res_ld,
        x.B, x.H, x.W, x.C, Ho, Wo, N, R, S, stride, pad_t, pad_l, upsample, dilate, accumulate, out_mode, splits, ws, cnt,
        x.dtype, _hip.stream()),
        f"conv M={x.B * Ho * Wo} N={N} K={R * S * x.C} {R}x{S} s{stride} u{upsample} d{dilate}",
        lambda: _hip.lib().ddpm_conv2d_variant(x.ld, y_ld, x.B, x.H, x.W, x.C, Ho, Wo, N, R, S, stride, pad_t, pad_l, upsample, dilate, out_mode, splits, x.dtype, (1 if res_ptr else 0) | (2 if accumulate else 0)))


_USE_SPLITK = bool(os.environ.get("DDPM_SPLITK"))
_SPLITK64 = os.environ.get("DDPM_SPLITK64", "1") != "0"


class SplitK:
    """Workspace for the in-launch split-K of layers with few output tiles: fp32 slabs + per-tile arrival counters (zero
    between launches: the last arriver of every tile resets its counter).  Launches that use it must be stream-ordered (the
    engine issues every conv forward / data gradient on the main stream).
    Default: TWO K runs per tile for the layers whose 64x64-tile grid has <= 128 blocks (the 4x4 level): the loop of that
    kernel is bound by what one CU can move into its LDS, so the second half of the chip halves it (scripts/g64_timeline.py;
    DDPM_SPLITK64=0 switches it off).  DDPM_SPLITK=1 additionally brings back the old 128x128-tile split-K for comparisons."""
    TARGET_BLOCKS = 128

    def __init__(self, device):
        self.device = device
        self.ws = None
        self.cnt = None

    # every plan of the small-grid kernel fits (2 / 4 / 8 runs, tiles x runs <= 256 slabs of 64 x 64): <= 128 tiles x 2 runs x one 128x128 fp32 tile.  Reserved in one piece at first use so that the
    # pointers a captured graph holds stay valid; a buffer that does get outgrown (DDPM_SPLITK=1 only) is retired, never freed.
    FLOATS64 = 128 * 2 * 16384

    def _reserve(self, floats, tiles):
        if self.ws is None or self.ws.numel() < floats:
            self._retired = getattr(self, "_retired", []) + [self.ws]
            self.ws = torch.empty(max(floats, self.FLOATS64), dtype=torch.float32, device=self.device)
        if self.cnt is None or self.cnt.numel() < tiles:
            self._retired = getattr(self, "_retired", []) + [self.cnt]
            self.cnt = torch.zeros(max(tiles, 1024), dtype=torch.int32, device=self.device)
        return self.ws.data_ptr(), self.cnt.data_ptr()

    def plan(self, M, N, K, dtype):
        ksteps = -(-K // (64 if dtype == _hip.BF16 else 32))
        tiles = -(-M // 128) * -(-N // 128)
        tiles64 = -(-M // 64) * -(-N // 64)
        if _SPLITK64 and tiles64 <= 128 and ksteps >= 16:
            # sized for either kernel the library may pick for these arguments (64x64 or 128x128 tiles)
            return (2,) + self._reserve(max(tiles64 * 2 * 4096, tiles * 2 * 16384), max(tiles64, tiles))
        if not _USE_SPLITK:
            return 1, 0, 0
        splits = min(self.TARGET_BLOCKS // tiles, ksteps // 8)
        if splits < 2:
            return 1, 0, 0
        return (splits,) + self._reserve(tiles * splits * 16384, tiles)


def conv2d_wgrad(dy, x, dw_ptr, Creal, Nreal, R, S, stride=1, pad_t=0, pad_l=0, upsample=0, splits=1, slab_stride=0):
    """slab_stride = 0: atomics into dw_ptr; > 0: split s stores its partial at dw_ptr + 4*s*slab_stride (see ddpm_wgrad_reduce)."""
    _timed("gemm_tt", 2.0 * dy.rows * Nreal * R * S * x.C, lambda: _hip.call(
        "ddpm_conv2d_wgrad_nhwc", dy.ptr, dy.ld, x.ptr, x.ld, dw_ptr, slab_stride, x.B, x.H, x.W, x.C, Creal, dy.H, dy.W, dy.C, Nreal, R, S,
        stride, pad_t, pad_l, upsample, splits, x.dtype, _hip.stream()), f"wgrad M={Nreal} N={R * S * x.C} K={dy.rows} splits={splits}",
        lambda: _hip.lib().ddpm_conv2d_wgrad_variant(dy.ld, x.ld, x.B, x.H, x.W, x.C, Creal, dy.H, dy.W, dy.C, Nreal, R, S, stride, pad_t, pad_l, upsample, splits, x.dtype))


def conv3x3_wgrad_splits(B, H, W, C, N, splits=0):
    """Slab copies the patch-stationary 3x3 wgrad kernel writes for this geometry; 0 = geometry not covered."""
    return int(_hip.lib().ddpm_conv3x3_wgrad_splits(B, H, W, C, N, splits))


def conv3x3_wgrad(dy, x, dw_ptr, slab_stride, dbias_ptr, bias_stride, Nreal, splits, upsample=False):
    """dw[n][3][3][c] (+ dbias) of a 3x3 / stride 1 / pad 1 conv by the patch-stationary kernel (bf16).  upsample: x is the input of an
    Upsample block, stored at half of dy's image size."""
    _timed("wgrad3x3", 2.0 * dy.rows * Nreal * 9 * x.C, lambda: _hip.call(
        "ddpm_conv3x3_wgrad_up_nhwc" if upsample else "ddpm_conv3x3_wgrad_nhwc", dy.ptr, dy.ld, x.ptr, x.ld, dw_ptr, slab_stride, dbias_ptr,
        bias_stride, dy.B, dy.H, dy.W, x.C, dy.C, Nreal, splits, x.dtype, _hip.stream()),
        f"wgrad3x3 M={Nreal} N={9 * x.C} K={dy.rows} splits={splits}" + (" up" if upsample else ""),
        lambda: max(6, int(_hip.lib().ddpm_conv3x3_wgrad_variant(dy.B, dy.H, dy.W, x.C, dy.C))))


def conv1x1_wgrad_splits(P, C, N):
    """Slab copies the 1x1 weight-gradient kernel writes for this geometry; 0 = not covered."""
    return int(_hip.lib().ddpm_conv1x1_wgrad_splits(P, C, N))


def conv1x1_wgrad(dy, x, dw_ptr, slab_stride, dbias_ptr, bias_stride, Nreal, splits):
    """dw[n][c] (+ dbias) of a 1x1 / stride 1 conv by the slab kernel (bf16)."""
    _timed("wgrad1x1", 2.0 * dy.rows * Nreal * x.C, lambda: _hip.call(
        "ddpm_conv1x1_wgrad_nhwc", dy.ptr, dy.ld, x.ptr, x.ld, dw_ptr, slab_stride, dbias_ptr, bias_stride, dy.rows, x.C, Nreal,
        splits, x.dtype, _hip.stream()), f"wgrad1x1 M={Nreal} N={x.C} K={dy.rows} splits={splits}", lambda: 9)


def wgrad_effective_splits(K, splits, dtype):
    return int(_hip.lib().ddpm_wgrad_effective_splits(K, splits, dtype))


def gemm(a_ptr, a_ld, a_bs, a_trans, b_ptr, b_ld, b_bs, b_trans, c_ptr, c_ld, c_bs, M, N, K, dtype, batch=1, alpha=1.0,
         bias=0, res_ptr=0, res_ld=0, res_bs=0, accumulate=0, out_mode=0, splits=1):
    _timed("gemm_" + "nt"[a_trans] + "nt"[b_trans], 2.0 * batch * M * N * K, lambda: _hip.call(
        "ddpm_gemm", a_ptr, a_ld, a_bs, a_trans, b_ptr, b_ld, b_bs, b_trans, c_ptr, c_ld, c_bs, bias, res_ptr, res_ld, res_bs,
        M, N, K, batch, alpha, accumulate, out_mode, splits, dtype, _hip.stream()), f"gemm b={batch} M={M} N={N} K={K} dt={dtype}",
        lambda: _hip.lib().ddpm_gemm_variant(a_ld, a_trans, b_ld, b_trans, c_ld, M, N, K, batch, out_mode, splits, dtype))


def gn_workspace_floats(B, HW, C, dtype):
    n = _hip.lib().ddpm_gn_workspace_floats(B, HW, C, GN_GROUPS, dtype)
    if n < 0:
        raise RuntimeError(f"GroupNorm geometry unsupported: B={B} HW={HW} C={C}")
    return n


def gn_fwd(x, y, gamma, beta, stats, ws, silu, drop_p=0.0, seed=0, seed_dev=0):
    """seed_dev: address of a device uint64 added to `seed` inside the kernel (0 = none) — the per-step part of the dropout
    seed when the step is a replayed hipGraph."""
    es = 2 if x.dtype == _hip.BF16 else 4
    # (bench.py's HBM roofline leg: algorithmic bytes = read x once + write y once)
    _timed("hbm_gn_fwd", 2.0 * x.B * x.H * x.W * x.C * es, lambda: _hip.call(
        "ddpm_groupnorm_silu_fwd", x.ptr, x.ld, y.ptr, y.ld, _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(stats), _hip.ptr(ws),
        x.B, x.H * x.W, x.C, GN_GROUPS, GN_EPS, int(silu), float(drop_p), seed, seed_dev, x.dtype, _hip.stream()),
        f"GroupNorm fwd B={x.B} HW={x.H * x.W} C={x.C}" + (" +dropout" if drop_p > 0 else ""))


def gn_bwd(x, dy, dx, gamma, beta, stats, dgamma_ptr, dbeta_ptr, ws, silu, drop_p=0.0, seed=0, accumulate=0, seed_dev=0, colsum_ptr=0, colsum_ld=0, add=None):
    """colsum_ptr: zero-initialised [B][colsum_ld] fp32 buffer that receives the per-sample channel sums of dx (0 = not wanted).
    add: View of a second gradient contribution summed into dx in the same pass (the identity branch of a residual connection)."""
    es = 2 if x.dtype == _hip.BF16 else 4
    # algorithmic bytes: read x and dy, write dx (+ read the identity-residual gradient it adds, + read dx when it accumulates)
    _timed("hbm_gn_bwd", (3.0 + (add is not None) + (1 if accumulate else 0)) * x.B * x.H * x.W * x.C * es, lambda: _hip.call(
        "ddpm_groupnorm_silu_bwd", x.ptr, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(stats), dgamma_ptr, dbeta_ptr,
        _hip.ptr(ws), x.B, x.H * x.W, x.C, GN_GROUPS, int(silu), float(drop_p), seed, seed_dev, accumulate, colsum_ptr, colsum_ld,
        add.ptr if add is not None else 0, add.ld if add is not None else 0, x.dtype, _hip.stream()),
        f"GroupNorm bwd B={x.B} HW={x.H * x.W} C={x.C}" + (" +dropout" if drop_p > 0 else "") + (" +add" if add is not None else ""))


def colsum(dy, per_sample_ptr, ps_ld, total_ptr):
    """per_sample[b][c] += sum_pixels dy;  total[c] += sum_{b,pixels} dy  (atomics into zero-initialised buffers)."""
    es = 2 if dy.dtype == _hip.BF16 else 4
    step = 64 * 256 * (16 // es)                 # one launch covers <= 64 groups of 256 16-byte channel vectors
    for c0 in range(0, dy.C, step):
        c1 = min(dy.C, c0 + step)
        _hip.call("ddpm_colsum", dy.ptr + c0 * es, dy.ld, per_sample_ptr + c0 * 4 if per_sample_ptr else 0, ps_ld,
             total_ptr + c0 * 4 if total_ptr else 0, dy.B, dy.H * dy.W, c1 - c0, dy.dtype, _hip.stream())


def resample2x(x, y, up, scale, accumulate=0):
    """2x resampling without a conv (csrc/elementwise.hip `resample2x_kernel`): up = 0 pools x [B, 2H, 2W] into y [B, H, W] (scale * 2x2 sum),
    up = 1 replicates x [B, H, W] into y [B, 2H, 2W]; (+)= when accumulate."""
    small = y if not up else x
    _hip.call("ddpm_resample2x_nhwc", x.ptr, x.ld, y.ptr, y.ld, small.B, small.H, small.W, small.C, up, scale, accumulate, x.dtype, _hip.stream())


def add_rows(x, y, accumulate):
    _hip.call("ddpm_add_rows", x.ptr, x.ld, y.ptr, y.ld, x.rows, x.C, accumulate, x.dtype, _hip.stream())
