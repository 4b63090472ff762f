"""Per-block phase timeline of the stationary-halo conv kernel (debug build with -DHALO_TIMING; timing only)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV = "cuda:0"
lib = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so"))
dt = torch.bfloat16
for (H, C, N) in ((32, 128, 128), (16, 256, 256), (16, 512, 256)):
    B, R = 128, 3
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, R * R * C, device=DEV) / math.sqrt(R * R * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    nblk = (B * H * H // 256) * (N // 128)
    tbuf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, R, R, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    assert lib.ddpm_debug_set_halo_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_halo_timing(ctypes.c_void_p(0))
    t = tbuf.view(nblk, 8).cpu().double()
    w0 = t[:, 0].min()
    start = (t[:, 0] - w0) / 100.0          # us (100 MHz wall clock)
    end = (t[:, 5] - w0) / 100.0
    pro, loop, epi = (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2]), (t[:, 4] - t[:, 3])
    stage = t[:, 6] - t[:, 3]
    tot_clk, tot_us = (t[:, 4] - t[:, 1]), end - start
    mhz = (tot_clk / tot_us).median()
    steps = 9 * C // 64
    print(f"H={H} C={C} N={N}: blocks={nblk} steps={steps} clk~{mhz:.0f} MHz  kernel span {end.max():.1f} us")
    print(f"  prologue {pro.median()/mhz:.2f} us  loop {loop.median()/mhz:.2f} us ({loop.median()/steps:.0f} clk/step)  epilogue {epi.median()/mhz:.2f} us (staging {stage.median()/mhz:.2f})  block total {tot_us.median():.2f} us")
    print(f"  block start times: first-round max {start.sort().values[min(255, nblk-1)]:.2f} us; second round starts {start.sort().values[min(256, nblk-1)]:.2f} .. {start.max():.2f} us")
