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


# ----------------------------------------------------------------------------------------------- Trainer.step at the bench batch
def _state_errors(actual, dg, lr_steps):
    """Sampled entries of every tensor against the fixture: the share beyond 1e-3 of the tensor's scale, the largest difference in units
    of lr x steps (Adam moves an element by at most lr per step whatever its gradient: a gradient at rounding-noise level flips sign
    between two implementations and the element ends up to 2 lr per step away), and the worst per-tensor error of the fp64 sums."""
    beyond = total = 0
    worst_lr, worst_sum = 0.0, 0.0
    for i, k in enumerate(dg["names"]):
        got, want = strided(actual[k], 64), dg["samples"][k]
        d = (got - want).abs()
        scale = max(float(want.abs().max()), 1e-3)
        beyond += int((d > 1e-3 * scale).sum()); total += d.numel()
        worst_lr = max(worst_lr, float(d.max()) / lr_steps)
        worst_sum = max(worst_sum, abs(float(actual[k].double().sum()) - float(dg["sum"][i])) / max(float(dg["abs_sum"][i]), 1e-30))
    return beyond / total, worst_lr, worst_sum


@pytest.mark.parametrize("dtype,form", [(torch.float32, False), (torch.bfloat16, False), (torch.float32, "plan"), (torch.bfloat16, "plan")])
def test_three_trainer_steps_at_batch_128_vs_reference_fixture(golden, monkeypatch, dtype, form):
    """utils/train.py:148-170 + EMA :300-305 by the reference's own Trainer (fixture G12) vs `ddpm_torch.Trainer.step` — the direct step with
    its side stream, slab reductions, fused clip + Adam + EMA — on the configs/cifar10.json network at B = 128, dropout 0, lr 1e-3 without
    warm-up (losses 2 and 3 depend on the weights written by the steps before), same CPU (t, noise) stream.  form "plan": step 2 records the
    launch plan, step 3 is a replay by csrc/plan.hip (the form the benchmark runs); form False: eager launches."""
    from ddpm_torch.utils import train as train_mod
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", form)
    g = golden("g12_config2_train_b128.pt")
    torch.manual_seed(g["init_seed"])
    m = ddpm_torch.UNet(**g["cfg"])
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), g["rand_seed"]))
    m = m.to(DEV).set_compute_dtype(dtype)
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=g["lr"], betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 1.0)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0, shape=(3, 32, 32),
                            device=torch.device(DEV), ema_decay=0.9999)
    gen = torch.Generator("cpu").manual_seed(g["gen_seed"])                  # the reference's CPU (t, noise) stream: utils/train.py:115,138-140

    def fill(t_buf, noise_buf):
        t_buf.copy_(torch.empty(t_buf.shape, dtype=torch.int64).random_(to=1000, generator=gen))
        noise_buf.copy_(torch.empty(noise_buf.shape).normal_(generator=gen))
    tr.input_source = fill
    m.train()
    losses = []
    for i, sd in enumerate(g["x_seeds"]):
        x = torch.rand(g["B"], 3, 32, 32, generator=torch.Generator().manual_seed(sd)) * 2 - 1
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    torch.cuda.synchronize()
    ds = next(iter(tr._direct.values()))
    assert ds.last_kind == ("plan" if form == "plan" else "eager")
    losses = torch.tensor(losses, dtype=torch.float64)
    rel = ((losses - g["losses"]).abs() / g["losses"]).tolist()
    steps = len(g["x_seeds"])
    pe = _state_errors({k: v.detach() for k, v in m.named_parameters()}, g["params"], g["lr"] * steps)
    se = _state_errors({k: tr.ema.shadow[k].detach() for k in g["shadow"]["names"]}, g["shadow"], g["lr"] * steps)
    print(f"G12 {dtype} B=128 {'launch plan' if form else 'eager'}: losses {[round(v, 5) for v in losses.tolist()]} vs {[round(v, 5) for v in g['losses'].tolist()]} (rel {['%.1e' % v for v in rel]}); "
          f"parameters: {pe[0]:.2%} of the sampled entries beyond 1e-3, worst {pe[1]:.2f} x lr x steps, sums {pe[2]:.1e}; EMA shadow: {se[0]:.2%}, {se[1]:.3f}, {se[2]:.1e}")
    assert tr.ema.num_updates == g["num_updates"]
    if dtype == torch.float32:
        assert rel[0] < 1e-5 and max(rel) < 2e-3
        assert pe[0] < 0.02 and pe[1] <= 0.75 and pe[2] < 1e-3
        assert se[0] < 0.02 and se[2] < 1e-3
    else:
        # lr 1e-3 from a randomised start is an AMPLIFYING regime (the loss climbs 1.64 -> 2.15 -> 2.84 in the reference too): the bf16 mode's
        # 1e-3 on the first loss grows to a few percent by the third.  Every element stays inside Adam's reach (<= 2 lr per step).
        # Bars = 2 x the measured 9.8e-4 / 3.7e-3 / 6.0e-2 (round 4 and 5 runs; the values are printed above).
        assert rel[0] < 2e-3 and rel[1] < 7.5e-3 and rel[2] < 1.2e-1
        assert pe[1] <= 2.0 and pe[2] < 5e-2
