"""Per-kernel timings on the GPU box (HIP events on the launch stream): the conv / GEMM shapes that carry the CIFAR
FLOPs and the fused GroupNorm+SiLU sites.  Prints one line per case + a JSON summary."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch  # noqa: E402

from ddpm_torch import _hip  # noqa: E402
from ddpm_torch import _ops as ops  # noqa: E402
from ddpm_torch._ops import View  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def conv_case(B, H, C, N, R, dtype, kind="fwd"):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dtype), B, H, H, C)
    w = (torch.randn(N, R * R * C, device=DEV) / math.sqrt(R * R * C)).to(dtype)
    y = View(torch.empty(B, H, H, N, device=DEV, dtype=dtype), B, H, H, N)
    bias = torch.zeros(N, device=DEV)
    flops = 2.0 * B * H * H * N * R * R * C
    if kind == "fwd":
        fn = lambda: ops.conv2d(x, w.data_ptr(), y.ptr, y.ld, N, R, R, H, H, pad_t=R // 2, pad_l=R // 2, bias=bias.data_ptr())
    else:
        dw = torch.zeros(N, C, R, R, device=DEV)
        tiles = -(-N // 128) * -(-(R * R * C) // 128)
        ksteps = B * H * H // (64 if dtype == torch.bfloat16 else 32)
        splits = max(1, min(512 // tiles, ksteps // (20 if ksteps >= 100 else 8)))      # the engine's rule (_Engine._splits)
        eff = ops.wgrad_effective_splits(B * H * H, splits, x.dtype)
        if os.environ.get("WGRAD_ATOMICS") or eff == 1:
            fn = lambda: ops.conv2d_wgrad(y, x, dw.data_ptr(), C, N, R, R, pad_t=R // 2, pad_l=R // 2, splits=eff)
        else:                                                     # product path: slab copies + one fixed-order reduction
            n = N * R * R * C
            slabs = torch.empty(eff * n, device=DEV)
            table = torch.tensor([[slabs.data_ptr(), dw.data_ptr(), n, eff, n]], dtype=torch.int64, device=DEV)

            def fn():
                ops.conv2d_wgrad(y, x, slabs.data_ptr(), C, N, R, R, pad_t=R // 2, pad_l=R // 2, splits=eff, slab_stride=n)
                _hip.call("ddpm_wgrad_reduce", table.data_ptr(), 1, _hip.stream())
    t = timeit(fn)
    return t, flops / t / 1e12


def attn_case(B, L, C):
    qkv = torch.randn(B * L, 3 * C, device=DEV).to(torch.bfloat16)
    o = torch.empty(B * L, C, device=DEV, dtype=torch.bfloat16)
    fn = lambda: _hip.call("ddpm_attention_fwd", qkv.data_ptr(), 3 * C, o.data_ptr(), C, B, L, C, 1.0 / math.sqrt(C), 1, _hip.stream())
    t = timeit(fn)
    return t, 4.0 * B * L * L * C / t / 1e12


def gn_case(B, H, C, dtype):
    x = View(torch.randn(B, H, H, C, device=DEV).to(dtype), B, H, H, C)
    y = View(torch.empty(B, H, H, C, device=DEV, dtype=dtype), B, H, H, C)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    ws = torch.empty(ops.gn_workspace_floats(B, H * H, C, x.dtype), device=DEV)
    fn = lambda: ops.gn_fwd(x, y, g, b, None, ws, True)
    t = timeit(fn)
    bytes_ = 2.0 * B * H * H * C * x.base.element_size()
    return t, bytes_ / t / 1e9


def main():
    out = {}
    shapes = [(128, 32, 128, 128, 3), (128, 16, 256, 256, 3), (128, 8, 256, 256, 3), (128, 4, 256, 256, 3), (128, 16, 512, 256, 3),
              (128, 32, 256, 128, 3), (128, 16, 256, 768, 1), (128, 16, 512, 256, 1)]
    for dtype in (torch.bfloat16, torch.float32):
        for (B, H, C, N, R) in shapes:
            for kind in ("fwd", "wgrad"):
                t, tf = conv_case(B, H, C, N, R, dtype, kind)
                key = f"conv{R}x{R}_{kind}_{str(dtype)[6:]}_B{B}_H{H}_C{C}_N{N}"
                out[key] = dict(ms=t * 1e3, tflops=tf)
                print(f"{key:60s} {t * 1e6:9.1f} us  {tf:8.1f} TFLOP/s", flush=True)
    for dtype in (torch.bfloat16, torch.float32):
        for (B, H, C) in [(128, 32, 128), (128, 16, 256), (128, 8, 256), (128, 4, 256), (128, 32, 384), (128, 16, 512)]:
            t, gbs = gn_case(B, H, C, dtype)
            key = f"gn_silu_fwd_{str(dtype)[6:]}_B{B}_H{H}_C{C}"
            out[key] = dict(ms=t * 1e3, gbps=gbs)
            print(f"{key:60s} {t * 1e6:9.1f} us  {gbs:8.1f} GB/s (algorithmic)", flush=True)
    for (B, L, C) in [(128, 256, 256), (128, 256, 128)]:
        t, tf = attn_case(B, L, C)
        key = f"attention_fused_fwd_bfloat16_B{B}_L{L}_C{C}"
        out[key] = dict(ms=t * 1e3, tflops=tf)
        print(f"{key:60s} {t * 1e6:9.1f} us  {tf:8.1f} TFLOP/s", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
