import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
def t(B, H, C, N, R, splits):
    dt = torch.bfloat16
    x = View(torch.randn(B, H, H, C, device='cuda').to(dt), B, H, H, C)
    y = View(torch.randn(B, H, H, N, device='cuda').to(dt), B, H, H, N)
    w = (torch.randn(N, R * R * C, device='cuda') * 0.02).to(dt)
    bias = torch.zeros(N, device='cuda')
    tiles = -(-(B * H * H) // 128) * -(-N // 128)
    ws = torch.empty(tiles * max(splits, 1) * 16384, device='cuda'); cnt = torch.zeros(tiles, dtype=torch.int32, device='cuda')
    f = lambda: _hip.call("ddpm_conv2d_nhwc", x.ptr, x.ld, w.data_ptr(), y.ptr, y.ld, bias.data_ptr(), 0, 0, 0, 0, B, H, H, C, H, H, N, R, R, 1, R // 2, R // 2, 0, 0, 0, 0,
                          splits, ws.data_ptr() if splits > 1 else 0, cnt.data_ptr() if splits > 1 else 0, x.dtype, _hip.stream())
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3
for s in (1, 2, 4, 8, 16):
    print(f"splits={s:2d}: 256->256@4x4 {t(128,4,256,256,3,s):6.1f} us | 256->256@8x8 {t(128,8,256,256,3,s):6.1f} us | 512->256@4x4 {t(128,4,512,256,3,s):6.1f} us | 1x1 512->256@4x4 {t(128,4,512,256,1,s):6.1f} us")
