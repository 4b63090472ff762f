"""3x3 conv forward timings on the CIFAR UNet's big shapes (B = 128, bf16): persistent stationary-halo kernel (conv3x3.hip) vs the
one-tile-per-block form (DDPM_CONV_NO_STREAM3=1 python scripts/c3_bench.py)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
tot = 0.0
for (H, C, N, cnt) in ((32, 128, 128, 14), (16, 256, 256, 12), (32, 256, 128, 2), (32, 128, 256, 2), (16, 512, 256, 2), (16, 256, 512, 2), (32, 384, 128, 1), (32, 128, 384, 1),
                       (16, 384, 256, 1), (16, 256, 384, 1), (16, 128, 256, 1), (16, 256, 128, 1)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    M = B * H * H
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 3, 3, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"H={H:2d} {C:3d}->{N:3d} x{cnt:2d}: {us:6.1f} us  {2.0 * M * N * 9 * C / us / 1e6:6.0f} TF", flush=True)
    tot += us * cnt
print(f"total {tot / 1e3:.3f} ms")
