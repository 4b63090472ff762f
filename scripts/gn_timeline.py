"""Per-block phase timeline of the LDS GroupNorm forward kernel (debug build with -DGN_TIMING, see gn_timeline.sh; timing only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
lib = ctypes.CDLL(_hip.LIB_PATH)          # (DDPM_HIP_LIB selects the instrumented build: scripts/gn_timeline.sh)
B = 128
def report(t, names):
    t = t[t[:, 0] > 0]
    w0 = t[:, 0].min()
    start, end = (t[:, 0] - w0) / 100.0, (t[:, 7] - w0) / 100.0
    mhz = ((t[:, 6] - t[:, 1]) / (end - start)).median()
    ph = [(t[:, i + 1] - t[:, i]).median() / mhz for i in range(1, 6)]
    print(f"   blocks={len(t)} clk~{mhz:.0f} MHz kernel span {end.max():.2f} us; block start spread {start.max():.2f} us; block total {((end - start).median()):.2f} us")
    print("   " + " | ".join(f"{n} {v:.2f}" for n, v in zip(names, ph)) + "  (us, medians)")
    # the same phases for the quarter of the blocks that END last (what the kernel's duration is made of), and where they run
    dur = end - start
    late = end >= end.quantile(0.75)
    phl = [(t[late, i + 1] - t[late, i]).median() / mhz for i in range(1, 6)]
    print("   last-ending quarter: " + " | ".join(f"{v:.2f}" for v in phl) + f"  (block total {float(dur[late].median()):.2f} us, start {float(start[late].median()):.2f} us)")
    q = lambda v: " ".join(f"{float(v.quantile(p)):.2f}" for p in (0.0, 0.25, 0.5, 0.75, 1.0))
    print("   block end quantiles (us):", q(end), " start quantiles:", q(start))


def capture(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tbuf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    assert lib.ddpm_debug_set_gn_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_gn_timing(ctypes.c_void_p(0))
    return tbuf.view(-1, 8).cpu().double()


for H, C in ((32, 128), (16, 256), (32, 256)):
    x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
    y, dy, dx, add = (View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C) for _ in range(4))
    g, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    stats = torch.zeros(B, 32, 2, device="cuda")
    cs = torch.zeros(B, C, device="cuda")
    ws = torch.zeros(ops.gn_workspace_floats(B, H * H, C, x.dtype), device="cuda")
    print(f"H={H} C={C} forward")
    report(capture(lambda: ops.gn_fwd(x, y, g, bt, stats, ws, silu=True, drop_p=0.1, seed=123)),
           ["load + moments pass", "wait for the slowest wave", "block reduction + group stats", "apply + store issue", "store drain"])
    print(f"H={H} C={C} backward (+ column sums of dx)")
    report(capture(lambda: ops.gn_bwd(x, dy, dx, g, bt, stats, dg.data_ptr(), db.data_ptr(), ws, silu=True, drop_p=0.1, seed=123, colsum_ptr=cs.data_ptr(), colsum_ld=C)),
           ["load + pass 1 (dz, sums)", "block reduction + coefficients", "pass 2 + store issue", "column sums", "store drain"])
