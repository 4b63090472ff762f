"""Phase timeline of the 64x64-tile kernel on the small levels (library built with -DHALO_TIMING): per-block prologue / loop / epilogue
and, for K-group 8, fragment reads + MFMA issue | LDS-DMA issue | vmcnt wait | barrier."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV = "cuda:0"
lib = ctypes.CDLL(os.environ.get("DDPM_HIP_LIB", os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so")))
dt = torch.bfloat16
B = int(os.environ.get("G64_B", "128"))
SK = ops.SplitK(DEV)          # DDPM_SPLITK64=0: one K run per tile
for (H, C, N, R) in ((4, 256, 256, 3), (4, 512, 256, 3), (4, 512, 256, 1), (4, 256, 768, 1), (8, 512, 256, 1), (8, 256, 256, 1)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, R * R * C, device=DEV) / math.sqrt(R * R * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    M = B * H * H
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, R, R, H, H, pad_t=R // 2, pad_l=R // 2, bias=bias.data_ptr(), splitk=SK)
    var = _hip.lib().ddpm_conv2d_variant(x.ld, y.ld, B, H, H, C, H, H, N, R, R, 1, R // 2, R // 2, 0, 0, 0, 1, x.dtype, 0)
    nblk = 8192
    tbuf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    assert lib.ddpm_debug_set_halo_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_halo_timing(ctypes.c_void_p(0))
    t = tbuf.view(nblk, 8).cpu()
    t = t[(t[:, 5] > 0) & (t[:, 4] > 0)]          # (blocks that only contributed a slab leave no end stamp)
    ph = t[:, 7]
    t = t.double()
    w0 = t[:, 0].min()
    start, end = (t[:, 0] - w0) / 100.0, (t[:, 5] - w0) / 100.0
    pro, loop, epi, stage = t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 6] - t[:, 3]
    mhz = ((t[:, 4] - t[:, 1]) / (end - start)).median()
    steps = R * R * C // 64
    q = [((ph >> s) & 0xffff).double().median().item() for s in (0, 16, 32, 48)]
    print(f"{R}x{R} H={H} C={C} N={N} variant {var}: {e0.elapsed_time(e1) * 50:.1f} us/launch blocks={len(t)} K-steps={steps} clk~{mhz:.0f} MHz span {end.max():.1f} us | "
          f"prologue {pro.median()/mhz:.2f}  loop {loop.median()/mhz:.2f} ({loop.median()/max(steps,1):.0f} clk/step)  epilogue {epi.median()/mhz:.2f} (staging {stage.median()/mhz:.2f}) "
          f"block {(end-start).median():.2f} us; start p50 {start.median():.1f} max {start.max():.1f} | group 8 clk: reads+mfma {q[0]:.0f} dma-issue {q[1]:.0f} vmcnt-wait {q[2]:.0f} barrier {q[3]:.0f}", flush=True)
