"""Per-block phase timeline + per-step clocks of conv3x3_pc_kernel's consumer wave 0 (library built with -DC3_TIMING; scripts/gpu_pc_tl.sh)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
lib = ctypes.CDLL(os.environ.get("DDPM_HIP_LIB") or os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so"))
ZERO = os.environ.get("ZERO_DATA") == "1"
for (H, C, N) in ((32, 128, 128), (16, 256, 256), (16, 512, 256)):
    xt = torch.randn(B, H, H, C, device=DEV).to(dt)
    if ZERO: xt.zero_()
    x = View(xt, B, H, H, C)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    if ZERO: w.zero_()
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 3, 3, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tbuf = torch.zeros(256 * 8 + 256 * 64, dtype=torch.int64, device=DEV)
    assert lib.ddpm_debug_set_c3_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_c3_timing(ctypes.c_void_p(0))
    t = tbuf[:2048].view(256, 8).cpu().double()
    st = tbuf[2048:].view(256, 64).cpu().double()
    ok = t[:, 7] > 0
    t, st = t[ok], st[ok]
    w0 = t[:, 0].min()
    start, end = (t[:, 0] - w0) / 100.0, (t[:, 7] - w0) / 100.0
    steps = 9 * C // 64
    two = bool((t[:, 6] > 0).all())
    last = 6 if two else 4
    mhz = ((t[:, last] - t[:, 1]) / (end - start)).median()
    d = lambda a, b: float((t[:, a] - t[:, b]).median() / mhz)
    msg = (f"H={H} {C}->{N}: blocks={len(t)} steps/tile={steps} clk~{mhz:.0f} MHz span {end.max():.1f} us start skew {start.max():.2f} | wait for prologue {d(2, 1):.2f}"
           f"  tile0 loop {d(3, 2):.2f} ({(t[:, 3] - t[:, 2]).median() / steps:.0f} clk/step)  epilogue {d(4, 3):.2f}")
    if two:
        msg += f"  tile1 loop {d(5, 4):.2f} ({(t[:, 5] - t[:, 4]).median() / steps:.0f} clk/step)  epilogue {d(6, 5):.2f}"
    print(msg + f"  block {(end - start).median():.2f} us", flush=True)
    n = min(64, steps * (2 if two else 1))
    dl = st[:, 1:n] - st[:, :n - 1]
    med = dl.median(dim=0).values
    print("   step deltas (clk, median over blocks): " + " ".join(f"{int(v)}" for v in med.tolist()))
    print("   block 0: " + " ".join(f"{int(v)}" for v in dl[0].tolist()))
