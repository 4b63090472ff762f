"""Time the attention training pair (ddpm_attention_fwd_lse + ddpm_attention_bwd) per geometry on the GPU.
    python scripts/attn_bench.py          # prints ms and MFMA TFLOP/s (4 B L^2 C forward, 10 B L^2 C backward incl. the recomputation)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ddpm-torch_amd"))
from ddpm_torch import _hip  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for B, L, C in [(128, 256, 256), (128, 64, 256), (128, 16, 256), (16, 256, 512), (128, 256, 128)]:
    qkv = torch.randn(B * L, 3 * C, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * L, C, device="cuda", dtype=torch.bfloat16)
    d_o = torch.randn(B * L, C, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    lse = torch.empty(B * L, device="cuda")
    dvec = torch.empty(B * L, device="cuda")
    sc = 1.0 / math.sqrt(C)
    st = _hip.stream()
    tf = timed(lambda: _hip.call("ddpm_attention_fwd_lse", qkv.data_ptr(), 3 * C, o.data_ptr(), C, lse.data_ptr(), B, L, C, sc, 1, st))
    tb = timed(lambda: _hip.call("ddpm_attention_bwd", qkv.data_ptr(), 3 * C, o.data_ptr(), C, d_o.data_ptr(), C, lse.data_ptr(), dvec.data_ptr(),
                                 dqkv.data_ptr(), 3 * C, B, L, C, sc, 1, st))
    line = f"B={B} L={L} C={C}: fwd_lse {tf:.3f} ms ({4 * B * L * L * C / tf / 1e9:.0f} TF)  bwd {tb:.3f} ms ({14 * B * L * L * C / tb / 1e9:.0f} TF incl. recompute)"
    if L % 128 == 0 and C in (128, 256):
        ti = timed(lambda: _hip.call("ddpm_attention_fwd", qkv.data_ptr(), 3 * C, o.data_ptr(), C, B, L, C, sc, 1, st))
        line += f"  inference fwd {ti:.3f} ms"
    print(line, flush=True)
