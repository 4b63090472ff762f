"""1x1-conv forward timings on the CIFAR UNet's shapes (B = 128, bf16): persistent streaming kernel vs the generic tile GEMM
(DDPM_CONV_NO_POINTWISE=1 python scripts/pw_bench.py for the latter)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
tot = 0.0
for (H, C, N, cnt, res) in ((32, 128, 256, 2, 0), (32, 256, 128, 2, 0), (32, 384, 128, 1, 0), (32, 128, 384, 1, 0), (16, 256, 256, 10, 1), (16, 256, 768, 5, 0),
                            (16, 768, 256, 5, 0), (16, 512, 256, 2, 0), (16, 256, 512, 2, 0), (16, 384, 256, 1, 0), (16, 256, 384, 1, 0)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, C, device=DEV) / math.sqrt(C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    r = View(torch.randn(B, H, H, N, device=DEV).to(dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    M = B * H * H
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 1, 1, H, H, bias=bias.data_ptr(), res_ptr=r.ptr if res else 0, res_ld=r.ld if res else 0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    mb = (M * C + M * N * (2 if res else 1)) * 2 / 1e6
    print(f"H={H:2d} {C:3d}->{N:3d} x{cnt:2d}{' +res' if res else '     '}: {us:6.1f} us  {2.0 * M * N * C / us / 1e6:6.0f} TF  {mb / us / 1e6 * 1e6 / 1e6:5.2f} TB/s", flush=True)
    tot += us * cnt
print(f"total {tot / 1e3:.3f} ms")
