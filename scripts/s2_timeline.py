"""Per-block timeline of the stride-2 data gradient on the generic GEMM kernel (debug build with -DHALO_TIMING; timing only):
   bash scripts/build_variant.sh timing -DHALO_TIMING gemm.hip && DDPM_HIP_LIB=.../libddpm_hip_timing.so python scripts/s2_timeline.py"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV = "cuda:0"; dt = torch.bfloat16; B = 128
lib = ctypes.CDLL(os.environ["DDPM_HIP_LIB"])
SK = ops.SplitK(DEV)

def report(name, fn, nblk):
    tbuf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    assert lib.ddpm_debug_set_halo_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_halo_timing(ctypes.c_void_p(0))
    t = tbuf.view(nblk, 8).cpu().double()
    t = t[t[:, 5] > 0]
    w0 = t[:, 0].min()
    start, end = (t[:, 0] - w0) / 100.0, (t[:, 5] - w0) / 100.0
    pro, loop, epi, stage = t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 6] - t[:, 3]
    mhz = ((t[:, 4] - t[:, 1]) / (end - start)).median()
    q = lambda v, p: float(torch.quantile(v, p))
    print(f"{name}: blocks={len(t)} clk~{mhz:.0f} MHz span {end.max():.1f} us | prologue p50 {pro.median()/mhz:.2f}  loop p10/p50/p90 {q(loop,.1)/mhz:.2f}/{q(loop,.5)/mhz:.2f}/{q(loop,.9)/mhz:.2f}"
          f"  epilogue p50 {epi.median()/mhz:.2f} p90 {q(epi,.9)/mhz:.2f} (staging {stage.median()/mhz:.2f})  block p50 {(end-start).median():.2f} p90 {q(end-start,.9):.2f} us; "
          f"starts p50 {start.median():.1f} p90 {q(start,.9):.1f} max {start.max():.1f}", flush=True)

for H, C in ((32, 128), (16, 256), (8, 256)):
    y = View(torch.randn(B, H // 2, H // 2, C, device=DEV).to(dt), B, H // 2, H // 2, C)
    w = (torch.randn(C, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    g = View(torch.empty(B, H, H, C, device=DEV, dtype=dt), B, H, H, C)
    fn = lambda: ops.conv2d(y, w.data_ptr(), g.ptr, g.ld, C, 3, 3, H, H, pad_t=2, pad_l=2, dilate=1, splitk=SK)
    report(f"dgrad s2 {H // 2}->{H} C={C}", fn, (B * H * H // 128) * (C // 128))
