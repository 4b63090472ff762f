"""GroupNorm streaming path on the CelebA-HQ tensors at the per-GPU batch of 2 (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
B = 2
for H, C in ((256, 128), (128, 128), (64, 256), (32, 256)):
    x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
    y, dy, dx = (View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C) for _ in range(3))
    g, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    stats = torch.zeros(B, 32, 2, device="cuda")
    ws = torch.zeros(ops.gn_workspace_floats(B, H * H, C, x.dtype), device="cuda")
    def timeit(fn, n=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    tf = timeit(lambda: ops.gn_fwd(x, y, g, bt, stats, ws, silu=True, drop_p=0.0, seed=0))
    tb = timeit(lambda: ops.gn_bwd(x, dy, dx, g, bt, stats, dg.data_ptr(), db.data_ptr(), ws, silu=True, drop_p=0.0, seed=0))
    print(f"B={B} {H}^2 x {C}: fwd {tf:6.1f} us  bwd {tb:6.1f} us", flush=True)
