"""Run ONE conv shape repeatedly (for rocprofv3 --pmc passes).  usage: one_kernel.py fwd|wgrad B H C N R [bf16|fp32] [iters]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _ops as ops
from ddpm_torch._ops import View
kind, B, H, C, N, R = sys.argv[1], *map(int, sys.argv[2:7])
dtype = torch.float32 if len(sys.argv) > 7 and sys.argv[7] == "fp32" else torch.bfloat16
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
x = View(torch.randn(B, H, H, C, device="cuda").to(dtype), B, H, H, C)
w = (torch.randn(N, R * R * C, device="cuda") / math.sqrt(R * R * C)).to(dtype)
y = View(torch.randn(B, H, H, N, device="cuda").to(dtype), B, H, H, N)
bias = torch.zeros(N, device="cuda")
dw = torch.zeros(N * C * R * R, device="cuda")
tiles = -(-N // 128) * -(-(R * R * C) // 128)
ksteps = B * H * H // (64 if dtype == torch.bfloat16 else 32)
splits = int(os.environ.get("SPLITS", max(1, min(ksteps // 4, -(-1024 // tiles)))))
for _ in range(iters):
    if kind == "fwd":
        ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, R, R, H, H, pad_t=R // 2, pad_l=R // 2, bias=bias.data_ptr())
    else:
        ops.conv2d_wgrad(y, x, dw.data_ptr(), C, N, R, R, pad_t=R // 2, pad_l=R // 2, splits=splits)
torch.cuda.synchronize()
