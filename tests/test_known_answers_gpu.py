"""-m gpu: known-answer checks of the HIP kernels that need neither the oracle nor the ABI emulator:
GroupNorm of a per-group constant -> beta, softmax rows sum to 1, attention with identical keys -> mean of V,
convolution with a centre-tap identity kernel -> identity (through all three forward conv kernels), stride-2 SAME padding."""
import math

import pytest
import torch

from ddpm_torch import _hip
from ddpm_torch import _ops as ops
from ddpm_torch._ops import View

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def test_groupnorm_of_a_groupwise_constant_is_beta():
    B, H, C = 4, 16, 256
    x = torch.zeros(B, H, H, C, device=DEV)
    x += torch.arange(32, device=DEV).repeat_interleave(C // 32).float() * 0.25 - 3.0      # constant inside every group
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    for dt in (torch.float32, BF):
        xv = View(x.to(dt).contiguous(), B, H, H, C)
        y = View(torch.empty(B, H, H, C, device=DEV, dtype=dt), B, H, H, C)
        ws = torch.empty(ops.gn_workspace_floats(B, H * H, C, xv.dtype), device=DEV)
        ops.gn_fwd(xv, y, gamma, beta, None, ws, False)
        torch.cuda.synchronize()
        ref = beta.to(dt).float().expand(B, H, H, C)
        # y = x * (rstd*gamma) + (beta - mean*rstd*gamma) with rstd = 1000: the two products cancel to ~1 ulp of |mean| * 1000
        assert float((y.base.float() - ref).abs().max()) <= (2e-3 if dt == torch.float32 else 3e-2)


def test_softmax_rows_sum_to_one():
    rows, L = 512, 256
    s = torch.randn(rows, L, device=DEV) * 4
    p = torch.empty(rows, L, device=DEV)
    _hip.call("ddpm_softmax_fwd", s.data_ptr(), p.data_ptr(), rows, L, _hip.F32, _hip.stream())
    torch.cuda.synchronize()
    assert float((p.sum(-1) - 1).abs().max()) < 1e-5 and float(p.min()) >= 0


def test_attention_with_identical_keys_returns_the_mean_of_v():
    B, L, C = 3, 256, 256
    qkv = torch.randn(B, L, 3 * C, device=DEV)
    qkv[:, :, C:2 * C] = qkv[:, :1, C:2 * C]                       # every key equal -> uniform attention
    qkv = qkv.to(BF).contiguous()
    out = torch.empty(B, L, C, device=DEV, dtype=BF)
    _hip.call("ddpm_attention_fwd", qkv.data_ptr(), 3 * C, out.data_ptr(), C, B, L, C, 1.0 / math.sqrt(C), _hip.BF16, _hip.stream())
    torch.cuda.synchronize()
    ref = qkv[:, :, 2 * C:].float().mean(1, keepdim=True).expand(B, L, C)
    assert float((out.float() - ref).abs().max()) < 2e-2


@pytest.mark.parametrize("B,H,C", [(16, 32, 128), (8, 8, 128), (2, 8, 64), (130, 16, 64)])
def test_conv3x3_with_centre_tap_identity_kernel_is_identity(B, H, C):
    """(16,32,128): stationary-halo kernel; (8,8,128): 128x128 generic; (2,8,64): 64x64 small-grid kernel; (130,16,64): ragged halo groups."""
    x = torch.randn(B, H, H, C, device=DEV).to(BF).contiguous()
    w = torch.zeros(C, 3, 3, C, device=DEV)
    w[:, 1, 1, :] = torch.eye(C, device=DEV)
    w = w.reshape(C, 9 * C).to(BF).contiguous()
    y = torch.empty(B, H, H, C, device=DEV, dtype=BF)
    ops.conv2d(View(x, B, H, H, C), w.data_ptr(), y.data_ptr(), C, C, 3, 3, H, H, pad_t=1, pad_l=1)
    torch.cuda.synchronize()
    assert torch.equal(y, x)


def test_stride2_same_padding_output_sizes():
    """SamePad2d(3, 2): 32 -> 16 with (top, left) = (0, 0) padding, 33 -> 17 with (1, 1) (modules.py:145-160)."""
    for H, Ho, pt in ((32, 16, 0), (33, 17, 1)):
        B, C = 2, 32
        x = torch.ones(B, H, H, C, device=DEV)
        w = torch.zeros(C, 3, 3, C, device=DEV)
        w[:, :, :, 0] = 1.0                                        # sums channel 0 over the 3x3 window
        w = w.reshape(C, 9 * C).contiguous()
        y = torch.empty(B, Ho, Ho, C, device=DEV)
        ops.conv2d(View(x, B, H, H, C), w.data_ptr(), y.data_ptr(), C, C, 3, 3, Ho, Ho, stride=2, pad_t=pt, pad_l=pt)
        torch.cuda.synchronize()
        # interior windows see 9 ones; a zero-padded border window sees fewer
        assert float(y[:, 1:-1, 1:-1].min()) == 9.0 and float(y.max()) == 9.0 and float(y.min()) in (4.0, 6.0)
        assert y.shape[1] == Ho
