"""Where a block of wgrad3x3_ws_kernel spends its time (library built with -DWG_TIMING: scripts/build_variant.sh wgt "-DWG_TIMING" wgrad.hip):
per wave the loop length, the clocks spent in semaphore waits (consumers: for the loaders' `ready`; loaders: for the consumers' `freed`), the
loaders' vmcnt waits, and the epilogue.   python scripts/wg_ws_timeline.py [lib suffix]"""
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
    t = tbuf.view(256, 8, 16).cpu().double()
    t = t[t[:, 0, 4] > 0]
    nst = int(t[0, 0, 4])
    loop, wait, vm, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2], t[:, :, 3], t[:, :, 5] - t[:, :, 1]
    print(f"H={H} {C}->{N}: copies={copies} blocks={t.shape[0]} stages/block={nst}  launch {us:.1f} us ({2.0 * B * H * H * N * 9 * C / us / 1e6:.0f} TFLOP/s)")
    print(f"   consumers (waves 0-3): loop {loop[:, :4].median():.0f} clk = {loop[:, :4].median() / nst:.0f} clk/stage (MFMA-bound: 4608); in `ready` waits {wait[:, :4].median():.0f} clk = {wait[:, :4].median() / nst:.0f} per stage")
    print(f"   loaders   (waves 4-7): loop {loop[:, 4:].median():.0f} clk; in `freed` waits {wait[:, 4:].median():.0f} = {wait[:, 4:].median() / nst:.0f} per stage; in vmcnt(0) {vm[:, 4:].median():.0f} = {vm[:, 4:].median() / nst:.0f} per stage")
    print(f"   after the loop (barrier + tile through LDS + stores): {epi.median():.0f} clk")
