"""Isolated timing of ddpm_atb_f32 on the training step's shapes (B = 128): the fc chunks' rows, embed lin2 / lin1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
K = 128
for M, N, lda in ((512, 512, 512), (512, 128, 512), (768, 512, 4992), (1280, 512, 4992), (4992, 512, 4992)):
    a = torch.randn(K, lda, device="cuda"); b = torch.randn(K, N, device="cuda"); c = torch.empty(M, N, device="cuda")
    fn = lambda: _hip.call("ddpm_atb_f32", a.data_ptr(), lda, b.data_ptr(), N, c.data_ptr(), N, M, N, K, _hip.stream())
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ref = a[:, :M].double().t() @ b.double()
    print(f"atb M={M} N={N} K={K}: {e0.elapsed_time(e1) * 20:.1f} us  max err {float((c.double() - ref).abs().max()):.2e}", flush=True)
