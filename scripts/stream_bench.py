"""Timings of the HBM-bound reduction kernels on the big tensors: column sums (bias gradients), GroupNorm forward (two-launch
path) and backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd"), os.path.join(ROOT, "scripts")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
from microbench import timeit
DEV = "cuda:0"; dt = torch.bfloat16
for (H, C) in ((32, 128), (32, 256), (16, 256), (16, 512)):
    B = 128
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    dy = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    dx = View(torch.empty(B, H, H, C, device=DEV, dtype=dt), B, H, H, C)
    y = View(torch.empty(B, H, H, C, device=DEV, dtype=dt), B, H, H, C)
    g, b_ = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    stats = torch.zeros(B, 32, 2, device=DEV); stats[..., 1] = 1
    ws = torch.empty(ops.gn_workspace_floats(B, H * H, C, x.dtype), device=DEV)
    tot = torch.zeros(C, device=DEV)
    mb = B * H * H * C * 2 / 1e6
    t_cs = timeit(lambda: ops.colsum(dy, 0, 0, tot.data_ptr()))
    t_f = timeit(lambda: ops.gn_fwd(x, y, g, b_, stats, ws, True))
    t_b = timeit(lambda: ops.gn_bwd(x, dy, dx, g, b_, stats, dg.data_ptr(), db.data_ptr(), ws, True))
    print(f"H={H} C={C} ({mb:.1f} MB): colsum {t_cs*1e6:.1f} us ({mb/t_cs/1e6:.2f} TB/s) | gn fwd {t_f*1e6:.1f} us ({2*mb/t_f/1e6:.2f} TB/s alg) | gn bwd {t_b*1e6:.1f} us ({3*mb/t_b/1e6:.2f} TB/s alg)")
