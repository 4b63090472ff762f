"""-m gpu: BASELINE config 2 at the BENCHMARK'S batch (B = 128) against the reference itself (fixture G11, written by
tests/golden/make_golden.py from the imported reference: upstream ddpm_torch/models/unet.py:205-233, ddpm_torch/diffusion.py:160-174).

This is the geometry at which the dispatcher picks the kernels the bench line is made of — the wave-specialised / persistent 3x3 kernels
(>= 16384 / 4096 pixels per layer), the streaming 1x1 kernel (>= 8192), the patch-stationary 3x3 and slab 1x1 weight gradients, the
flash attention forward / backward, LDS GroupNorm — so here they meet reference-written numbers directly: an eval-mode forward, EVERY
parameter gradient of sum(y * gy), and an 8-step ancestral chain through the graph-replayed sampler.  The fixture keeps strided samples and
fp64 sums of the tensors (35.7 M gradients do not travel), inputs are seeds.

fp32 mode: <= 1e-3 of the tensor's range.  bf16 mode (what the bench measures): stated, measured bars, printed."""
import pytest
import torch

import ddpm_torch
from ddpm_torch import _hip
from oracle import unet_ref as U
from tests.golden.recipes import rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g11(golden):
    return golden("g11_config2_b128.pt")


def strided(t, n=256):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).round().long().to(f.device)
    return f[idx].float().cpu()


def shipped(rec, dtype, train):
    cfg = dict(rec["cfg"], drop_rate=0.0)                       # the reference ran in eval mode: no dropout on either side
    torch.manual_seed(rec["init_seed"])
    m = ddpm_torch.UNet(**cfg)
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), rec["rand_seed"]))
    m = m.to(DEV).set_compute_dtype(dtype)
    return m.train() if train else m.eval()


def test_this_batch_runs_the_hot_kernels(g11):
    """The point of the fixture: at B = 128 the layers of configs/cifar10.json are dispatched to the kernels of the bench line."""
    lib, B = _hip.lib(), g11["B"]
    conv = lambda H, C, N, R, ep=0: lib.ddpm_conv2d_variant(C, N, B, H, H, C, H, H, N, R, R, 1, R // 2, R // 2, 0, 0, 0, 1, 1, ep)
    assert conv(32, 128, 128, 3) == 13 and conv(16, 256, 256, 3) == 13 and conv(16, 512, 256, 3) == 13     # conv3x3_pc_kernel
    assert conv(32, 128, 128, 3, 1) == 8                                                                     # residual epilogue: conv3x3_stream_kernel<16>
    assert conv(8, 256, 256, 3) == 10                                                                        # conv3x3_stream_kernel<8>
    assert conv(16, 256, 256, 1) == 7 and conv(16, 256, 768, 1) == 7 and conv(8, 512, 256, 1) == 7          # pw_conv_kernel
    assert lib.ddpm_conv3x3_wgrad_splits(B, 32, 32, 128, 128, 0) > 0 and lib.ddpm_conv1x1_wgrad_splits(B * 256, 256, 768) > 0


def _forward_backward(g11, dtype):
    m = shipped(g11, dtype, train=True)
    B, f, g = g11["B"], g11["fwd"], g11["grads"]
    x, gy = rnd(B, 3, 32, 32, seed=f["x_seed"]), rnd(B, 3, 32, 32, seed=g["gy_seed"])
    y = m(x.to(DEV), f["t"].to(DEV))
    (y * gy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    yd = y.detach().float().cpu()
    scale = f["y_absmax"]
    fwd_err = float((yd[:, :, ::4, ::4] - f["y_sub"]).abs().max()) / scale
    sum_err = float(((yd.double().sum((1, 2, 3)) - f["y_sum"]).abs() / f["y_abs"]).max())
    rows = []
    params = dict(m.named_parameters())
    assert list(params) == g["names"]
    norms = g["sq_sum"].sqrt()
    med = float(norms.median())
    for i, k in enumerate(g["names"]):
        got, want = strided(params[k].grad), g["samples"][k]
        # error of the sampled entries relative to the tensor's largest sampled gradient (floor: a tenth of the median tensor's RMS
        # entry, for the tensors whose gradient is analytically ~0)
        floor = 0.1 * med / max(params[k].numel(), 1) ** 0.5
        e = float((got - want).abs().max()) / max(float(want.abs().max()), floor)
        s = abs(float(params[k].grad.double().sum()) - float(g["sum"][i])) / max(float(g["abs_sum"][i]), 1e-30)
        rows.append((e, s, k))
    return fwd_err, sum_err, sorted(rows, reverse=True)


def test_forward_and_every_gradient_fp32_vs_reference_fixture(g11):
    fwd_err, sum_err, rows = _forward_backward(g11, torch.float32)
    # measured: forward 2.6e-6, per-image sums 6.3e-7, worst gradient tensor 1.7e-5 (sampled entries) / 1.8e-6 (sum)
    print(f"G11 fp32 B=128: forward max err / range {fwd_err:.3e}, per-image sums {sum_err:.3e}; worst gradient tensors (sampled entries, sum): "
          + ", ".join(f"{k} {e:.2e}/{s:.2e}" for e, s, k in rows[:4]))
    assert fwd_err < 1e-3 and sum_err < 1e-4
    assert rows[0][0] < 3e-3, rows[:4]                            # weight gradients add split-K slices in another order than the reference's GEMMs
    assert max(s for _, s, _ in rows) < 1e-3


def test_forward_and_every_gradient_bf16_with_stated_bars(g11):
    """The bench's own mode, i.e. conv3x3_pc / conv3x3_stream / pw_conv / wgrad3x3 / wgrad1x1 / flash attention / LDS GroupNorm against
    reference-written numbers.  Bars: forward max < 4e-2 of the range (measured 1.5e-2); every gradient tensor's sampled entries within
    7e-2 of the tensor's largest sampled gradient (measured: worst 3.5e-2), the median tensor within 3e-2 (measured 1.8e-2) — bf16
    activations and packed weights under fp32 accumulation: the class tests/test_configs_gpu.py measures against the oracle at B = 4."""
    fwd_err, sum_err, rows = _forward_backward(g11, torch.bfloat16)
    errs = sorted(e for e, _, _ in rows)
    print(f"G11 bf16 B=128: forward max err / range {fwd_err:.3e}; gradient tensors: median {errs[len(errs) // 2]:.2e}, worst "
          + ", ".join(f"{k} {e:.2e}" for e, _, k in rows[:4]))
    assert fwd_err < 4e-2
    assert errs[len(errs) // 2] < 3e-2 and rows[0][0] < 7e-2, rows[:4]


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])        # measured: 2.1e-6 / 1.5e-2
def test_eight_step_chain_at_batch_128_through_the_captured_sampler(g11, monkeypatch, dtype, bar):
    """diffusion.py:160-174 at B = 128 on the reference's CPU noise stream, through the graph-replayed sampler."""
    m = shipped(g11, dtype, train=False)
    r = g11["ddpm8_fixed-large"]
    shape = tuple(r["shape"])
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, r["timesteps"]), "eps", "fixed-large", "mse")
    g = torch.Generator("cpu").manual_seed(r["seed"])
    x_T = torch.empty(shape).normal_(generator=g)
    zs = [torch.empty(shape).normal_(generator=g) for _ in range(r["timesteps"])]
    monkeypatch.setenv("DDPM_TORCH_AMD_GRAPH", "1")
    with torch.inference_mode():
        x = dif._sample_loop(m, shape, DEV, x_T, None, z_stream=iter(zs))
    assert any(k[1] == shape and k[3][1] for k in dif._sample_graphs), "the chain did not run through the captured step"
    xd = x.detach().float().cpu()
    e = float((xd[:, :, ::4, ::4] - r["x0_sub"]).abs().max()) / r["x0_absmax"]
    s = float(((xd.double().sum((1, 2, 3)) - r["x0_sum"]).abs() / r["x0_abs"]).max())
    print(f"G11 8-step chain B=128 {dtype}: max err / range {e:.3e}, per-image sums {s:.3e}")
    assert torch.isfinite(xd).all() and e < bar
