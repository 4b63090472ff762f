"""Isolated timing of the layers that still run on the generic GEMM kernel in the CIFAR step: stride-2 downsample convs (forward, dgrad as a
dilated conv), in_conv / out_conv (forward, dgrad).  B = 128, bf16."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV = "cuda:0"; dt = torch.bfloat16; B = 128
SK = ops.SplitK(DEV)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def V(B, H, W, C): return View(torch.randn(B, H, W, C, device=DEV).to(dt), B, H, W, C)

for H, C in ((32, 128), (16, 256), (8, 256)):
    x, y = V(B, H, H, C), V(B, H // 2, H // 2, C)
    w = (torch.randn(C, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    bias = torch.zeros(C, device=DEV)
    fwd = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, C, 3, 3, y.H, y.W, stride=2, pad_t=0, pad_l=0, bias=bias.data_ptr(), splitk=SK)
    g = V(B, H, H, C)
    dgrad = lambda: ops.conv2d(y, w.data_ptr(), g.ptr, g.ld, C, 3, 3, H, H, pad_t=2, pad_l=2, dilate=1, splitk=SK)
    gf = 2 * B * (H // 2) ** 2 * 9 * C * C / 1e9
    tf, tb = timeit(fwd), timeit(dgrad)
    print(f"down {H}->{H // 2} C={C}: fwd {tf:6.1f} us ({gf / tf * 1e-3:6.1f} TF)   dgrad {tb:6.1f} us ({gf / tb * 1e-3:6.1f} TF useful)", flush=True)
# in_conv 3 -> 128 (input padded to 8 channels), out_conv 128 -> 3
for name, Cin, Cout in (("in_conv", 8, 128), ("out_conv", 128, 8)):
    x, y = V(B, 32, 32, Cin), V(B, 32, 32, Cout)
    w = (torch.randn(Cout, 9 * Cin, device=DEV) / math.sqrt(9 * Cin)).to(dt)
    bias = torch.zeros(Cout, device=DEV)
    if name == "out_conv":        # as the engine calls it: 3 real output channels, NCHW fp32 result
        w = (torch.randn(3, 9 * Cin, device=DEV) / math.sqrt(9 * Cin)).to(dt)
        o32 = torch.empty(B, 3, 32, 32, device=DEV)
        fwd = lambda: ops.conv2d(x, w.data_ptr(), o32.data_ptr(), 0, 3, 3, 3, 32, 32, pad_t=1, pad_l=1, bias=bias.data_ptr(), out_mode=3, splitk=SK)
    else:
        fwd = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, Cout, 3, 3, 32, 32, pad_t=1, pad_l=1, bias=bias.data_ptr(), splitk=SK)
    print(f"{name}: fwd {timeit(fwd):6.1f} us", flush=True)
    if name == "out_conv":
        wd = (torch.randn(Cin, 9 * Cout, device=DEV)).to(dt)
        dg = lambda: ops.conv2d(y, wd.data_ptr(), x.ptr, x.ld, Cin, 3, 3, 32, 32, pad_t=1, pad_l=1, splitk=SK)
        print(f"{name}: dgrad {timeit(dg):6.1f} us", flush=True)
