"""Per-shape kernel durations of the 1x1 convs via the kernel trace: one shape per process run is ambiguous in the stats, so each shape
is launched a distinctive number of times (printed) and the trace CSV is post-processed by the caller."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
H, C, N, res = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
w = (torch.randn(N, C, device=DEV) / math.sqrt(C)).to(dt)
y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
r = View(torch.randn(B, H, H, N, device=DEV).to(dt), B, H, H, N)
bias = torch.zeros(N, device=DEV)
for _ in range(30):
    ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 1, 1, H, H, bias=bias.data_ptr(), res_ptr=r.ptr if res else 0, res_ld=r.ld if res else 0)
    torch.cuda.synchronize()
