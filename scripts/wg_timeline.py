"""Phase timeline of ONE mid-kernel stage of wgrad3x3_kernel<256, 16, 3> per wave (library built with -DWG_TIMING:
scripts/build_variant.sh wgt "-DWG_TIMING" wgrad.hip).  Stamps sit right behind full waits.  Usage: python scripts/wg_timeline.py [lib suffix]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
DEV, dt, B = "cuda:0", torch.bfloat16, 128
lib = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", f"libddpm_hip_{sys.argv[1] if len(sys.argv) > 1 else 'wgt'}.so"))
for name in ("ddpm_conv3x3_wgrad_nhwc", "ddpm_conv3x3_wgrad_splits"):
    getattr(lib, name).argtypes = _hip.PROTOTYPES[name]; getattr(lib, name).restype = ctypes.c_int
st = torch.cuda.current_stream().cuda_stream
NAMES = ["MFMA f4 (6) + issue f5,f0 reads", "wait f5,f0", "MFMA f5,f0 (6) + issue f1", "wait f1", "MFMA f1 (6) + issue f2", "wait f2", "MFMA f2 (9) + issue f3", "wait f3",
         "wait_vm (next stage landed)", "barrier", "issue next A,f4 + MFMA f3 (9)", "DMA issue", "wait A,f4 + copies"]
for (H, C, N) in ((32, 128, 128), (16, 256, 256), (32, 256, 128)):
    x = torch.randn(B, H, H, C, device=DEV).to(dt)
    dy = torch.randn(B, H, H, N, device=DEV).to(dt)
    n = N * 9 * C
    copies = lib.ddpm_conv3x3_wgrad_splits(B, H, H, C, N, 0)
    slab = torch.empty(copies * (n + N), device=DEV)
    fn = lambda: lib.ddpm_conv3x3_wgrad_nhwc(dy.data_ptr(), N, x.data_ptr(), C, slab.data_ptr(), n, slab.data_ptr() + 4 * copies * n, N, B, H, H, C, N, N, 0, 1, st)
    for _ in range(3): assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    tbuf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=DEV)
    assert lib.ddpm_debug_set_wg_timing(ctypes.c_void_p(tbuf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    lib.ddpm_debug_set_wg_timing(ctypes.c_void_p(0))
    t = tbuf.view(256, 8, 16).cpu()
    t = t[t[:, 0, 14] > 0]                               # blocks that ran
    nst = int(t[0, 0, 14] >> 32)
    t[:, :, 14] &= 0xffffffff
    d = ((t[:, :, 1:13] - t[:, :, 0:12]) & 0xffffffff).double()
    d = torch.cat([d, ((t[:, :, 15:16] - t[:, :, 12:13]) & 0xffffffff).double()], dim=2)
    total = ((t[:, :, 14] - t[:, :, 13]) & 0xffffffff).double()
    print(f"H={H} {C}->{N}: splits(copies)={copies} blocks={t.shape[0]} stages/block={nst}  launch {us:.1f} us ({2.0 * B * H * H * N * 9 * C / us / 1e6:.0f} TFLOP/s)  "
          f"loop {total.median():.0f} clk = {total.median() / nst:.0f} clk/stage (MFMA-bound: 2304 with two waves per SIMD)")
    med = d.median(dim=0).values                          # [wave][phase]
    for k in range(13):
        print(f"   {NAMES[k]:38s} " + " ".join(f"{int(v):5d}" for v in med[:, k].tolist()) + f"   | all waves {d[:, :, k].median():6.0f}")
    print(f"   {'sum of the stamped stage':38s} " + " ".join(f"{int(v):5d}" for v in med.sum(1).tolist()))
