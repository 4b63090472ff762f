"""1x1-conv launches of the CIFAR step through the C ABI (no Python wrapper in the timed loop), bursts of N launches on random data:
plain / +residual / "+=" epilogues, and the kernel's timing-only ablations (DDPM_PW_ABLATE: 1 = no stores, 2 = no MFMA, 3 = both) —
where does a 1x1 launch spend its time?  Usage: python scripts/pw_ab.py [launches per burst]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
DEV, dt, B = "cuda:0", torch.bfloat16, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
h = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
# (H, C_in, C_out, launches per step, epilogue) — epilogue: 0 plain, 1 residual, 2 "+=" (how the step calls them)
SHAPES = ((32, 128, 256, 2, 2), (32, 128, 384, 1, 2), (32, 256, 128, 2, 0), (32, 384, 128, 1, 0), (16, 256, 256, 10, 1), (16, 256, 768, 5, 0),
          (16, 768, 256, 5, 0), (16, 256, 512, 2, 2), (16, 512, 256, 2, 0), (16, 256, 384, 1, 2), (16, 384, 256, 1, 0), (8, 512, 256, 3, 0), (8, 256, 512, 3, 2))
tot = {}
for (H, C, N, cnt, ep) in SHAPES:
    x = torch.randn(B, H, H, C, device=DEV).to(dt)
    w = (torch.randn(N, C, device=DEV) / math.sqrt(C)).to(dt)
    y = torch.zeros(B, H, H, N, device=DEV, dtype=dt)
    r = torch.randn(B, H, H, N, device=DEV).to(dt)
    bias = torch.zeros(N, device=DEV)
    M = B * H * H

    def fn(mode):
        rc = h.ddpm_conv2d_nhwc(x.data_ptr(), C, w.data_ptr(), y.data_ptr(), N, bias.data_ptr(), 0, 0, r.data_ptr() if mode == 1 else 0, N if mode == 1 else 0,
                                B, H, H, C, H, H, N, 1, 1, 1, 0, 0, 0, 0, 1 if mode == 2 else 0, 0, 1, 0, 0, 1, st)
        assert rc == 0, rc

    def burst(mode):
        fn(mode); torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn(mode)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        return best
    line = f"H={H:2d} {C:3d}->{N:3d} x{cnt:2d}:"
    for mode, name in ((0, "plain"), (1, "+res"), (2, "+=")):
        os.environ["DDPM_PW_ABLATE"] = "0"
        y.zero_()
        t = burst(mode)
        mb = (M * C + M * N * (2 if mode else 1)) * 2 / 1e6
        line += f"  {name} {t:6.1f} us {mb / t:4.2f} TB/s"
        if mode == ep: tot["prod"] = tot.get("prod", 0.0) + t * cnt
        if mode == 0: tot["plain"] = tot.get("plain", 0.0) + t * cnt
    for ab, name in ((1, "no-stores"), (2, "no-mfma"), (3, "neither")):
        os.environ["DDPM_PW_ABLATE"] = str(ab)
        line += f"  | {name} {burst(0):6.1f}"
    os.environ["DDPM_PW_ABLATE"] = "0"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y2 = torch.empty_like(y)
    e0.record()
    for _ in range(50): y2.copy_(y)
    e1.record(); torch.cuda.synchronize()
    line += f"  | copy of y {e0.elapsed_time(e1) * 20:5.1f} us"
    print(line, flush=True)
print("per step: " + "  ".join(f"{k} {v / 1e3:.3f} ms" for k, v in tot.items()))
