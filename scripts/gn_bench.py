"""Isolated timing of the GroupNorm(+SiLU+dropout) forward / backward launches on the CIFAR UNet's tensor shapes (B = 128, bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
B = 128
ACC = int(os.environ.get('GN_ACC', '0'))          # 1: the backward adds into dx ("+=")
DROP = float(os.environ.get('GN_DROP', '0.1')); SILU = os.environ.get('GN_SILU', '1') == '1'
SHAPES = [(32, 128, 8), (32, 256, 2), (32, 384, 1), (16, 256, 11), (16, 512, 2), (16, 384, 1), (16, 128, 1), (8, 256, 7), (8, 512, 3), (4, 256, 12), (4, 512, 3)]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot_f = tot_b = 0.0
if os.environ.get('GN_ONLY'):
    SHAPES = SHAPES[:int(os.environ['GN_ONLY'])]
for H, C, cnt in SHAPES:
    x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
    y, dy, dx = (View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C) for _ in range(3))
    g, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    stats = torch.zeros(B, 32, 2, device="cuda")
    cs = torch.zeros(B, C, device="cuda")
    ws = torch.zeros(ops.gn_workspace_floats(B, H * H, C, x.dtype), device="cuda")
    tf = timeit(lambda: ops.gn_fwd(x, y, g, bt, stats, ws, silu=SILU, drop_p=DROP, seed=123))
    tb = timeit(lambda: ops.gn_bwd(x, dy, dx, g, bt, stats, dg.data_ptr(), db.data_ptr(), ws, silu=SILU, drop_p=DROP, seed=123, colsum_ptr=cs.data_ptr(), colsum_ld=C, accumulate=ACC))
    mb = B * H * H * C * 2 / 1e6
    tc = timeit(lambda: y.base.copy_(x.base))
    print(f"{H:2d}^2 x {C:3d} x{cnt:2d}: fwd {tf:6.1f} us ({2 * mb / tf:5.2f} TB/s)  bwd {tb:6.1f} us ({3 * mb / tb:5.2f} TB/s)   [copy_ {tc:5.1f} us {2 * mb / tc:5.2f} TB/s]", flush=True)
    tot_f += tf * cnt; tot_b += tb * cnt
print(f"network totals: fwd {tot_f / 1e3:.3f} ms  bwd {tot_b / 1e3:.3f} ms")
