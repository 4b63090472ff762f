"""3x3 conv forward timings on the 8x8 level (B = 128, bf16): 8x8-patch persistent kernel vs the 64x64-tile GEMM (DDPM_CONV_NO_STREAM3_8=1)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
for (H, C, N, cnt) in ((8, 256, 256, 14), (8, 512, 256, 3), (8, 256, 512, 1)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 3, 3, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"H={H} {C:3d}->{N:3d} x{cnt:2d}: {us:6.1f} us  {2.0 * B * H * H * N * 9 * C / us / 1e6:6.0f} TF", flush=True)
