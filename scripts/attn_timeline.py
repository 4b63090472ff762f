"""Phase timeline of attn_fwd_lse_kernel (library built with -DATTN_TIMING): medians over the blocks, 100 MHz wall clock."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
lib = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", "libddpm_hip.so"))
for B, L, C in ((128, 256, 256), (128, 256, 128), (128, 64, 256)):
    qkv = torch.randn(B * L, 3 * C, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * L, C, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B * L, device="cuda")
    fn = lambda: _hip.call("ddpm_attention_fwd_lse", qkv.data_ptr(), 3 * C, o.data_ptr(), C, lse.data_ptr(), B, L, C, 1.0 / math.sqrt(C), 1, _hip.stream())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tb = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    assert lib.ddpm_debug_set_attn_timing(ctypes.c_void_p(tb.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_attn_timing(ctypes.c_void_p(0))
    t = tb.view(-1, 8).cpu().double(); t = t[t[:, 0] > 0]
    w0 = t[:, 0].min()
    names = ["P1 (Q K^T)", "softmax", "write tile", "P2 (P V)", "store issue", "store drain"]
    ph = [((t[:, i + 1] - t[:, i]) / 100.0).median().item() for i in range(6)]
    q7 = tb.view(-1, 8).cpu()[:, 7]; q7 = q7[q7 != 0]
    if len(q7):
        q = [((q7 >> sft) & 0xffff).double().median().item() for sft in (0, 16, 32, 48)]
        print(f"   P1 chunk 1, wave 0 (clk): vmcnt wait {q[0]:.0f} | barrier {q[1]:.0f} | DMA issue {q[2]:.0f} | reads + MFMA {q[3]:.0f}")
    print(f"B={B} L={L} C={C}: blocks={len(t)} span {((t[:, 6].max() - w0) / 100.0):.1f} us; start spread {((t[:, 0].max() - w0) / 100):.1f} us | " +
          " | ".join(f"{n} {v:.2f}" for n, v in zip(names, ph)), flush=True)
