"""Sustained rate of the 3x3 weight-gradient kernel alone (direct C-ABI launches, no Python wrapper in the loop): wgrad_sustained.py [launches]
Run with DDPM_WGRAD3_CUS=256 / 128 to see the full-chip and the half-chip (product) forms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lib = _hip.lib()
B, DEV = 128, "cuda:0"
st = torch.cuda.current_stream().cuda_stream
for (H, C, N) in ((32, 128, 128), (16, 256, 256), (32, 256, 128), (16, 512, 256)):
    x = torch.randn(B, H, H, C, device=DEV).bfloat16()
    dy = torch.randn(B, H, H, N, device=DEV).bfloat16()
    nw = N * 9 * C
    copies = int(lib.ddpm_conv3x3_wgrad_splits(B, H, H, C, N, 0))
    slab = torch.empty(copies * (nw + N), device=DEV)
    def fn():
        rc = lib.ddpm_conv3x3_wgrad_nhwc(dy.data_ptr(), N, x.data_ptr(), C, slab.data_ptr(), nw, slab.data_ptr() + 4 * copies * nw, N, B, H, H, C, N, N, 0, 1, st)
        assert rc == 0, rc
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / n)
    us = min(ts)
    fl = 2.0 * B * H * H * N * 9 * C
    print(f"wgrad3x3 H={H} {C}->{N}: copies={copies}  {us:7.1f} us  {fl / us / 1e6:6.0f} TF  ({fl / us / 1e6 / 2500:.3f} of nominal peak)  [CUS={os.environ.get('DDPM_WGRAD3_CUS', '128')}]", flush=True)
