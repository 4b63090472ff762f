"""Fixed cost vs per-K-step cost of the conv kernel on the small-M (8x8 / 4x4) levels: sweep C (K-steps) and split-K."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
from microbench import timeit

DEV = "cuda:0"
dt = torch.bfloat16
FAST = bool(os.environ.get("SWEEP_FAST"))
for H in ((8, 16) if FAST else (4, 8, 16)):
    for C in ((128, 512) if FAST else (64, 128, 256, 512)):
        B, N, R = 128, 256, 3
        x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
        w = (torch.randn(N, R * R * C, device=DEV) / math.sqrt(R * R * C)).to(dt)
        y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
        bias = torch.zeros(N, device=DEV)
        tiles = (B * H * H // 128) * (N // 128)
        ws = torch.empty(tiles * 16 * 16384, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(4096, dtype=torch.int32, device=DEV)
        out = []
        for splits in ((1,) if FAST else (1, 2, 4, 8, 16)):
            if splits > 1 and (R * R * C // 64) // splits < 2: continue
            fn = lambda: _hip.call("ddpm_conv2d_nhwc", x.ptr, x.ld, w.data_ptr(), y.ptr, y.ld, bias.data_ptr(), 0, 0, 0, 0,
                                   B, H, H, C, H, H, N, R, R, 1, 1, 1, 0, 0, 0, 0, splits, ws.data_ptr() if splits > 1 else 0,
                                   cnt.data_ptr() if splits > 1 else 0, x.dtype, _hip.stream())
            t = timeit(fn, iters=50)
            out.append(f"s{splits}:{t * 1e6:6.1f}")
        print(f"H={H:2d} C={C:3d} ksteps={R * R * C // 64:3d} tiles={tiles:4d}  " + "  ".join(out), flush=True)
