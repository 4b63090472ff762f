for s in 8 16 28 56 114 228; do SPLITS=$s python - <<PY
import os, sys, math, torch
sys.path[:0]=['.', 'ddpm-torch_amd']
from ddpm_torch import _ops as ops
from ddpm_torch._ops import View
def t(B,H,C,N,R,splits):
    dt=torch.bfloat16
    x=View(torch.randn(B,H,H,C,device='cuda').to(dt),B,H,H,C); y=View(torch.randn(B,H,H,N,device='cuda').to(dt),B,H,H,N)
    dw=torch.zeros(N*C*R*R,device='cuda')
    f=lambda: ops.conv2d_wgrad(y,x,dw.data_ptr(),C,N,R,R,pad_t=R//2,pad_l=R//2,splits=splits)
    for _ in range(3): f()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/20*1e3
s=int(os.environ['SPLITS'])
print(s, "128->128@32: %.1f us | 256->256@16: %.1f us | 256->256@8: %.1f us | 256->256@4: %.1f us" % (t(128,32,128,128,3,s), t(128,16,256,256,3,s), t(128,8,256,256,3,s), t(128,4,256,256,3,s)))
PY
done
