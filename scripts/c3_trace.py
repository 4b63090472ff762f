"""Per-step phase trace of the persistent 3x3 kernel (library built with -DC3_TRACE): waves 0 and 4 of block 0, clock64 stamps."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV, dt, B = "cuda:0", torch.bfloat16, 128
FIRST = int(os.environ.get("FIRST", "5"))
lib = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so"))
names = ["sub0", "dma", "wait", "sub1", "dma", "wait", "sub2", "dma", "wait", "sub3", "vmwait", "lgkm", "barrier"]
for (H, C, N) in ((32, 128, 128), (16, 256, 256)):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dt), B, H, H, C)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dt), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, 3, 3, H, H, pad_t=1, pad_l=1, bias=bias.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tbuf = torch.zeros(2 * 64 * 16, dtype=torch.int64, device=DEV)
    assert lib.ddpm_debug_set_c3_trace(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_c3_trace(ctypes.c_void_p(0))
    t = tbuf.view(2, 64, 16).cpu()
    print(f"H={H} {C}->{N}: clk since the step's first stamp of wave 0; phases: " + " ".join(names))
    for q in range(4):
        for wv in (0, 1):
            row = (t[wv, q, :14] - t[0, q, 0]).tolist()
            d = [row[i + 1] - row[i] for i in range(13)]
            print(f"  step {FIRST + q} wave {4 * wv}: start {row[0]:5d} end {row[13]:5d} | " + " ".join(f"{n}={v}" for n, v in zip(names, d)))
