import math, os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd"), os.path.join(ROOT, "scripts")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
from microbench import timeit
DEV = "cuda:0"; dt = torch.bfloat16
for (H, C, N) in ((32, 128, 128), (16, 256, 256), (32, 256, 128), (16, 512, 256)):
    B = 128
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    a = View(torch.empty(B, H, H, C, device=DEV, dtype=dt), B, H, H, C)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    g, b_ = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    bias = torch.zeros(N, device=DEV)
    stats = torch.zeros(B, 32, 2, device=DEV)
    ws = torch.empty(ops.gn_workspace_floats(B, H * H, C, x.dtype), device=DEV)
    t_stats = timeit(lambda: ops.gn_stats(x, stats))
    t_fold = timeit(lambda: ops.conv3x3_gn(x, stats, g, b_, w.data_ptr(), y.ptr, y.ld, N, bias=bias.data_ptr()))
    t_gn = timeit(lambda: ops.gn_fwd(x, a, g, b_, None, ws, True))
    t_conv = timeit(lambda: ops.conv2d(a, w.data_ptr(), y.ptr, y.ld, N, 3, 3, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr()))
    print(f"H={H} C={C} N={N}: stats {t_stats*1e6:.1f} + folded conv {t_fold*1e6:.1f} = {(t_stats+t_fold)*1e6:.1f} us | gn_fwd {t_gn*1e6:.1f} + conv {t_conv*1e6:.1f} = {(t_gn+t_conv)*1e6:.1f} us")
