"""1x1 weight-gradient timings on the CIFAR UNet's shapes (B = 128, bf16): slab kernel (csrc/wgrad1x1.hip) vs the generic
both-operands-k-strided GEMM with split-K atomics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_new = tot_old = 0.0
for (H, C, N, cnt) in ((16, 256, 768, 5), (16, 256, 256, 5), (16, 512, 256, 1), (16, 384, 256, 1), (32, 256, 128, 1), (32, 384, 128, 1), (16, 128, 256, 1)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    dy = View(torch.randn(B, H, H, N, device=DEV).to(dt), B, H, H, N)
    P = B * H * H
    sp = ops.conv1x1_wgrad_splits(P, C, N)
    slab = torch.empty(sp * (N * C + N), device=DEV)
    dw = torch.zeros(N * C, device=DEV)
    t_new = timeit(lambda: ops.conv1x1_wgrad(dy, x, slab.data_ptr(), N * C, slab.data_ptr() + 4 * sp * N * C, N, N, sp))
    tiles = -(-N // 128) * -(-C // 128)
    ks = P // 64
    splits = ops.wgrad_effective_splits(P, max(1, min(512 // tiles, ks // 20)), x.dtype)
    t_old = timeit(lambda: ops.conv2d_wgrad(dy, x, dw.data_ptr(), C, N, 1, 1, splits=splits))
    fl = 2.0 * P * N * C
    print(f"H={H:2d} dW[{N:3d}][{C:3d}] x{cnt}: slab kernel {t_new:6.1f} us ({fl / t_new / 1e6:5.0f} TF, {sp} copies)   generic {t_old:6.1f} us ({fl / t_old / 1e6:5.0f} TF, {splits} slices)", flush=True)
    tot_new += t_new * cnt; tot_old += t_old * cnt
print(f"totals: {tot_new / 1e3:.3f} ms vs {tot_old / 1e3:.3f} ms")
