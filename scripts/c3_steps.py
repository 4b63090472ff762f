"""Per-step clock stamps of block 0, tile 0 of the persistent 3x3 kernel (library built with -DC3_TIMING -DC3_STEPTIMING)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
lib = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so"))
for (H, C, N) in ((32, 128, 128), (16, 256, 256)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 3, 3, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tbuf = torch.zeros(4096, dtype=torch.int64, device=DEV)
    assert lib.ddpm_debug_set_c3_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_c3_timing(ctypes.c_void_p(0))
    t = tbuf.cpu()
    steps = 9 * C // 64
    st = t[2048:2048 + steps].double()
    d = (st[1:] - st[:-1]).tolist()
    print(f"H={H} {C}->{N}: prologue->first step end {float(st[0] - t[2]):.0f} clk; step deltas:", " ".join(f"{v:.0f}" for v in d))
