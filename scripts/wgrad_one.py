"""One shape of the patch-stationary wgrad kernel, a few launches (for rocprofv3 --pmc runs): H C N [splits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _ops as ops
from ddpm_torch._ops import View
H, C, N = (int(v) for v in sys.argv[1:4]); spl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
B = 128
x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
dy = View(torch.randn(B, H, H, N, device="cuda").bfloat16(), B, H, H, N)
n = N * 9 * C
copies = ops.conv3x3_wgrad_splits(B, H, H, C, N, spl)
slab = torch.empty(copies * (n + N), device="cuda")
for _ in range(5):
    ops.conv3x3_wgrad(dy, x, slab.data_ptr(), n, slab.data_ptr() + 4 * copies * n, N, N, spl)
torch.cuda.synchronize()
