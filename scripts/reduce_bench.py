"""wgrad_reduce in isolation: six 3x3 conv-weight gradients (the flush size of the backward) with the slab counts of the CIFAR layers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
DEV = "cuda:0"
for n, copies, cnt in ((128 * 9 * 128, 32, 6), (256 * 9 * 256, 8, 6), (256 * 9 * 512, 4, 3)):
    stride = (n + 3) // 4 * 4
    slabs = [torch.randn(copies * stride, device=DEV) for _ in range(cnt)]
    dst = [torch.empty(n, device=DEV) for _ in range(cnt)]
    table = torch.tensor([[s.data_ptr(), d.data_ptr(), n, copies, stride] for s, d in zip(slabs, dst)], dtype=torch.int64, device=DEV)
    fn = lambda: _hip.call("ddpm_wgrad_reduce", table.data_ptr(), cnt, _hip.stream())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    mb = cnt * (copies + 1) * n * 4 / 1e6
    ref = slabs[0].view(copies, stride)[:, :n].double().sum(0)
    err = float((dst[0].double() - ref).abs().max() / ref.abs().max())
    print(f"{cnt} tensors x {n} floats x {copies} copies: {us:6.1f} us, {mb:6.1f} MB -> {mb / us / 1e6 * 1e6 / 1e3:5.2f} TB/s   rel err {err:.1e}", flush=True)
