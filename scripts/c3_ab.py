"""Same-process A/B of the 3x3 conv kernels of two builds of the library (libddpm_hip_prev.so vs libddpm_hip.so): interleaved bursts of
launches on random data, so that clock / thermal drift hits both alike.  Usage: python scripts/c3_ab.py [launches per burst]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
DEV, dt, B = "cuda:0", torch.bfloat16, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
libs = {}
for tag, name in (("prev", "libddpm_hip_prev.so"), ("new", "libddpm_hip.so")):
    h = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", name))
    h.ddpm_conv2d_nhwc.argtypes = _hip.PROTOTYPES["ddpm_conv2d_nhwc"]
    h.ddpm_conv2d_nhwc.restype = ctypes.c_int
    libs[tag] = h
st = torch.cuda.current_stream().cuda_stream
for (H, C, N, res) in ((32, 128, 128, 0), (16, 256, 256, 0), (32, 128, 128, 1), (16, 256, 256, 1), (32, 256, 128, 0), (16, 512, 256, 0), (8, 256, 256, 0)):
    x = torch.randn(B, H, H, C, device=DEV).to(dt)
    w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
    y = torch.empty(B, H, H, N, device=DEV, dtype=dt)
    r = torch.randn(B, H, H, N, device=DEV).to(dt)
    bias = torch.zeros(N, device=DEV)
    def fn(h):
        rc = h.ddpm_conv2d_nhwc(x.data_ptr(), C, w.data_ptr(), y.data_ptr(), N, bias.data_ptr(), 0, 0, r.data_ptr() if res else 0, N if res else 0,
                                B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 1, st)
        assert rc == 0, rc
    out = {}
    for tag, h in libs.items():
        fn(h); torch.cuda.synchronize(); out[tag] = y.clone()
    same = torch.equal(out["prev"], out["new"])
    t = {"prev": [], "new": []}
    for rep in range(4):
        for tag, h in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn(h)
            e1.record(); torch.cuda.synchronize()
            t[tag].append(e0.elapsed_time(e1) * 1e3 / n)
    fl = 2.0 * B * H * H * N * 9 * C
    p, q = min(t["prev"][1:]), min(t["new"][1:])
    print(f"H={H:2d} {C:3d}->{N:3d} res={res}: prev {p:6.1f} us {fl / p / 1e6:5.0f} TF | new {q:6.1f} us {fl / q / 1e6:5.0f} TF | {100 * (p / q - 1):+5.1f} %  bit-identical={same}   bursts prev {['%.1f' % v for v in t['prev']]} new {['%.1f' % v for v in t['new']]}", flush=True)
