"""Isolated timing of the 3x3 weight-gradient kernels on the CIFAR UNet's layer shapes (B = 128, bf16):
patch-stationary kernel (slab / atomic, library-chosen or forced slice counts) vs the generic transposed-operand GEMM.
    python scripts/wgrad_bench.py [splits ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View

SHAPES = [  # H, Cin, Cout, count in the network
    (32, 128, 128, 7), (32, 256, 128, 2), (32, 384, 128, 1), (32, 256, 256, 1),
    (16, 256, 256, 7), (16, 512, 256, 2), (16, 384, 256, 1), (16, 128, 256, 1),
    (8, 256, 256, 9), (8, 512, 256, 3), (4, 256, 256, 12), (4, 512, 256, 3),
]
B = 128


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def main():
    forced = [int(v) for v in sys.argv[1:]]
    tot = {}
    for H, C, N, cnt in SHAPES:
        x = View(torch.randn(B, H, H, C, device="cuda").bfloat16(), B, H, H, C)
        dy = View(torch.randn(B, H, H, N, device="cuda").bfloat16(), B, H, H, N)
        flops = 2.0 * B * H * H * N * 9 * C
        n = N * 9 * C
        row = f"{H:2d}^2 {C:3d}->{N:3d} x{cnt:2d} |"
        # generic
        dw = torch.zeros(n, device="cuda")
        tiles = -(-N // 128) * -(-(9 * C) // 128)
        ks = -(-(B * H * H) // 64)
        sp = max(1, min(512 // tiles, ks // (20 if ks >= 100 else 8)))
        eff = ops.wgrad_effective_splits(B * H * H, sp, x.dtype)
        t = timeit(lambda: ops.conv2d_wgrad(dy, x, dw.data_ptr(), C, N, 3, 3, pad_t=1, pad_l=1, splits=eff))
        row += f" generic {t:6.1f} us {flops / t / 1e6:6.0f} TF |"
        tot["generic"] = tot.get("generic", 0) + t * cnt
        for spl in [0] + forced:
            copies = ops.conv3x3_wgrad_splits(B, H, H, C, N, spl)
            stride = (n + 3) // 4 * 4
            slab = torch.empty(copies * (stride + N), device="cuda")
            bias = slab.data_ptr() + 4 * copies * stride
            t = timeit(lambda: ops.conv3x3_wgrad(dy, x, slab.data_ptr(), stride, bias, N, N, spl))
            ta = timeit(lambda: ops.conv3x3_wgrad(dy, x, dw.data_ptr(), 0, 0, 0, N, spl))
            row += f" s={copies:3d}: slab {t:6.1f} us {flops / t / 1e6:6.0f} TF, atomic {ta:6.1f} us |"
            tot[f"slab{spl}"] = tot.get(f"slab{spl}", 0) + t * cnt
            tot[f"atomic{spl}"] = tot.get(f"atomic{spl}", 0) + ta * cnt
        print(row, flush=True)
    print("network totals (ms):", {k: round(v / 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
