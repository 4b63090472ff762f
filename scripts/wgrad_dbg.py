"""Per-wave cycle breakdown of the patch wgrad kernel's main loop (debug instantiation ABL = 32): H C N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
H, C, N = (int(v) for v in sys.argv[1:4])
B = 128
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["DDPM_WG_DBGPTR"] = str(dbg.data_ptr()); os.environ["DDPM_WG_ABLATE"] = sys.argv[4] if len(sys.argv) > 4 else "32"
from ddpm_torch import _ops as ops
from ddpm_torch._ops import View
x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
dy = View(torch.randn(B, H, H, N, device="cuda").bfloat16(), B, H, H, N)
n = N * 9 * C
copies = ops.conv3x3_wgrad_splits(B, H, H, C, N, 0)
slab = torch.empty(copies * (n + N), device="cuda")
for _ in range(3):
    ops.conv3x3_wgrad(dy, x, slab.data_ptr(), n, slab.data_ptr() + 4 * copies * n, N, N, 0)
torch.cuda.synchronize()
nb = (N // 64) * (C // 32) * copies
d = dbg[:nb * 32].reshape(nb, 8, 4).double().cpu()
print("blocks", nb, "stages/block", B * (H // 16) ** 2 // copies)
print("mean per wave: loop %.0f  barrier %.0f  lgkm %.0f  dma-wait %.0f cycles" % tuple(d.mean((0, 1)).tolist()))
print("by wave (loop):", [int(v) for v in d[:, :, 0].mean(0).tolist()])
print("by wave (barrier):", [int(v) for v in d[:, :, 1].mean(0).tolist()])
print("by wave (lgkm):", [int(v) for v in d[:, :, 2].mean(0).tolist()])
print("by wave (dma):", [int(v) for v in d[:, :, 3].mean(0).tolist()])
