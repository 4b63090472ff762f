"""Per-block phase timeline of the LDS GroupNorm forward kernel (debug build with -DGN_TIMING, see gn_timeline.sh; timing only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
lib = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so"))
B = 128
for H, C, nblk in ((32, 128, 512), (16, 256, 512), (32, 256, 1024)):
    x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
    y = View(torch.empty(B, H, H, C, device="cuda", dtype=torch.bfloat16), B, H, H, C)
    g, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    stats = torch.zeros(B, 32, 2, device="cuda")
    ws = torch.zeros(ops.gn_workspace_floats(B, H * H, C, x.dtype), device="cuda")
    fn = lambda: ops.gn_fwd(x, y, g, bt, stats, ws, silu=True, drop_p=0.1, seed=123)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tbuf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    assert lib.ddpm_debug_set_gn_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_gn_timing(ctypes.c_void_p(0))
    t = tbuf.view(-1, 8).cpu().double()
    t = t[t[:, 0] > 0]
    w0 = t[:, 0].min()
    start, end = (t[:, 0] - w0) / 100.0, (t[:, 7] - w0) / 100.0
    tot_clk = t[:, 6] - t[:, 1]
    mhz = (tot_clk / (end - start)).median()
    ph = [(t[:, i + 1] - t[:, i]).median() / mhz for i in range(1, 6)]
    print(f"H={H} C={C}: blocks={len(t)} clk~{mhz:.0f} MHz kernel span {end.max():.2f} us; block start spread {start.max():.2f} us; block total {((end - start).median()):.2f} us")
    print("   load + moments pass %.2f | wait for the slowest wave %.2f | block reduction + group stats %.2f | apply + store issue %.2f | store drain %.2f  (us, medians)" % tuple(ph))
    q = lambda v: " ".join(f"{float(v.quantile(p)):.2f}" for p in (0.0, 0.25, 0.5, 0.75, 1.0))
    print("   block end quantiles (us):", q(end), " start quantiles:", q(start))
