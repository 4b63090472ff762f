"""-m gpu: does the headline mode TRAIN?  (upstream ddpm_torch/utils/train.py:148-170 `Trainer.step`, :300-305 EMA)

(a) 300 `Trainer.step`s of the CIFAR-10 configuration (configs/cifar10.json: dropout 0.1, Adam, clip 1.0, warm-up, EMA) at B = 128 on a
    fixed 256-image synthetic set, once in the bf16 throughput mode and once in the fp32 parity mode — same initial weights, same
    CPU (t, noise) stream (the reference's generator, utils/train.py:115,138-140), same dropout seeds.  The two loss curves must
    fall (measured 0.93 -> 0.097 fp32 / 0.095 bf16), and stay within a stated band of each other (12 % per 25-step window; measured
    <= 6.2 %); the EMA shadows' distance is reported.
(b) 50 fp32 steps of a mid-size net (hid 64, 16 x 16, attention at 8 x 8) against oracle/train_ref.py on the same stream, with a
    learning rate that moves the weights: per-step losses and the final parameters.

The bands are measured values with margin; the measurements are printed."""
import math

import pytest
import torch

import ddpm_torch
from oracle import diffusion_ref as D
from oracle import train_ref
from oracle import unet_ref as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CIFAR = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
             apply_attn=[False, True, False, False], drop_rate=0.1)
MID = dict(in_channels=3, hid_channels=64, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=2, apply_attn=[False, True], drop_rate=0.0)


def synthetic_images(n, hw, seed):
    """Smooth, structured images in [-1, 1] (a few random plane waves per channel through tanh): something a denoiser can learn."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, hw), torch.linspace(0, 1, hw), indexing="ij")
    img = torch.zeros(n, 3, hw, hw)
    for _ in range(4):
        f = torch.randn(n, 3, 2, generator=g) * 3.0
        ph = torch.rand(n, 3, 1, 1, generator=g) * 2 * math.pi
        amp = torch.randn(n, 3, 1, 1, generator=g)
        img += amp * torch.cos(2 * math.pi * (f[..., 0, None, None] * yy + f[..., 1, None, None] * xx) + ph)
    return torch.tanh(img)


def reference_stream(seed=8191):
    gen = torch.Generator("cpu").manual_seed(seed)

    def fill(t_buf, noise_buf):
        t_buf.copy_(torch.empty(t_buf.shape, dtype=torch.int64).random_(to=1000, generator=gen))
        noise_buf.copy_(torch.empty(noise_buf.shape).normal_(generator=gen))
    return fill


def run(cfg, dtype, data, batch, steps, lr, warmup, init_seed=1234):
    torch.manual_seed(init_seed)
    m = ddpm_torch.UNet(**cfg).to(DEV).set_compute_dtype(dtype)
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: min((s + 1) / warmup, 1.0))
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0, shape=tuple(data.shape[1:]),
                            device=torch.device(DEV), ema_decay=0.9999)
    tr.input_source = reference_stream()
    m.train()
    losses = []
    nb = data.shape[0] // batch
    for i in range(steps):
        tr.stats.reset()
        tr.step(data[(i % nb) * batch:(i % nb + 1) * batch], global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    torch.cuda.synchronize()
    return m, tr, torch.tensor(losses, dtype=torch.float64)


def test_bf16_mode_trains_like_the_fp32_mode():
    data = synthetic_images(256, 32, seed=3)
    steps, win = 300, 25
    m16, tr16, l16 = run(CIFAR, torch.bfloat16, data, 128, steps, lr=2e-4, warmup=50)
    m32, tr32, l32 = run(CIFAR, torch.float32, data, 128, steps, lr=2e-4, warmup=50)
    w16, w32 = l16.reshape(-1, win).mean(1), l32.reshape(-1, win).mean(1)
    print("loss (mean of 25 steps)  fp32: " + " ".join(f"{v:.4f}" for v in w32.tolist()))
    print("loss (mean of 25 steps)  bf16: " + " ".join(f"{v:.4f}" for v in w16.tolist()))
    gap = ((w16 - w32).abs() / w32).max()
    step_gap = ((l16 - l32).abs() / l32)
    sh = {k: (tr16.ema.shadow[k].float() - tr32.ema.shadow[k].float()).norm() / tr32.ema.shadow[k].float().norm().clamp_min(1e-12) for k in tr32.ema.shadow}
    num = math.sqrt(sum(float((tr16.ema.shadow[k].float() - tr32.ema.shadow[k].float()).norm()) ** 2 for k in sh))
    den = math.sqrt(sum(float(tr32.ema.shadow[k].float().norm()) ** 2 for k in sh))
    moved = math.sqrt(sum(float((p.detach().float().cpu() - q).norm()) ** 2 for p, q in zip(m32.parameters(), _initial(CIFAR)))) / den
    print(f"windowed-loss gap bf16 vs fp32: max {float(gap):.3%}; per-step gap: median {float(step_gap.median()):.3%}, max {float(step_gap.max()):.3%}; "
          f"EMA shadow distance {num / den:.3e} of its norm (the fp32 weights moved {moved:.3e} of it from the initial point)")
    assert torch.isfinite(l16).all() and torch.isfinite(l32).all()
    assert w32[-1] < 0.35 * w32[0] and w16[-1] < 0.35 * w16[0], "the loss did not fall"          # trains: measured ~0.1 x
    assert float(gap) < 0.12, "bf16 and fp32 loss curves drifted apart"                         # measured: <= 6.2 % in every 25-step window (0.93 -> 0.097 / 0.095)
    # (no bar on the weights themselves: two Adam trajectories that differ by rounding separate in parameter space at the rate they
    #  move — measured: the bf16 and fp32 shadows end 6.7e-2 of their norm apart after moving 6.5e-2 from the start — while reaching the
    #  same loss; the reference run against itself at another thread count does the same)


def _initial(cfg, init_seed=1234):
    torch.manual_seed(init_seed)
    return [p.detach().clone() for p in ddpm_torch.UNet(**cfg).parameters()]


def test_fp32_steps_track_the_oracle_on_a_mid_size_net():
    data = synthetic_images(32, 16, seed=4)
    steps, B, lr, warmup = 50, 8, 1e-3, 10
    m, tr, losses = run(MID, torch.float32, data, B, steps, lr=lr, warmup=warmup, init_seed=77)
    torch.manual_seed(77)
    sd0 = U.init_state_dict(MID)
    st = train_ref.TrainState(sd0, MID, lr=lr, warmup=warmup, grad_norm=1.0, ema_decay=0.9999)
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-large")
    gen = torch.Generator("cpu").manual_seed(8191)
    ref = []
    nb = data.shape[0] // B
    for i in range(steps):
        x = data[(i % nb) * B:(i % nb + 1) * B]
        t = torch.empty((B,), dtype=torch.int64).random_(to=1000, generator=gen)
        noise = torch.empty_like(x).normal_(generator=gen)
        ref.append(st.step(T, x, t, noise, training=True))
    ref = torch.tensor(ref, dtype=torch.float64)
    rel = ((losses - ref).abs() / ref)
    print(f"50 fp32 steps vs oracle: loss {float(ref[0]):.4f} -> {float(ref[-1]):.4f}; per-step relative gap: median {float(rel.median()):.2e}, max {float(rel.max()):.2e}")
    assert float(rel[:10].max()) < 2e-3 and float(rel.max()) < 3e-2          # early steps tight; Adam amplifies rounding differences later
    num = math.sqrt(sum(float((p.detach().cpu() - st.params[k].detach()).norm()) ** 2 for k, p in m.named_parameters()))
    den = math.sqrt(sum(float((st.params[k].detach() - sd0[k]).norm()) ** 2 for k in sd0))
    print(f"parameter distance to the oracle after 50 steps: {num / den:.3e} of the distance travelled")
    assert num / den < 0.15
