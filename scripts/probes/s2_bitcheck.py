import math, os, sys, hashlib
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip, _ops as ops
from ddpm_torch._ops import View
DEV="cuda:0"; dt=torch.bfloat16; B=128
SK = ops.SplitK(DEV)
torch.manual_seed(0)
for H, C in ((32,128),(16,256),(8,256)):
    y = View(torch.randn(B,H//2,H//2,C,device=DEV).to(dt),B,H//2,H//2,C)
    w = (torch.randn(C,9*C,device=DEV)/math.sqrt(9*C)).to(dt)
    g = View(torch.randn(B,H,H,C,device=DEV).to(dt),B,H,H,C)
    ops.conv2d(y,w.data_ptr(),g.ptr,g.ld,C,3,3,H,H,pad_t=2,pad_l=2,dilate=1,accumulate=1,splitk=SK)
    torch.cuda.synchronize()
    print(H, C, hashlib.sha1(g.base.cpu().view(torch.int16).numpy().tobytes()).hexdigest())
