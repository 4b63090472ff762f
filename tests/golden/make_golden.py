"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):  ``python tests/golden/make_golden.py``.
It imports the upstream modules through ``oracle/load_reference.py`` (stub package, no
torchvision), runs them on seeded CPU inputs and saves inputs + expected outputs as small
``.pt`` files (plain dicts of tensors / python scalars, loadable with ``weights_only=True``).
No reference source travels: fixtures are data only.  Coverage follows SURVEY.md §8c G1-G8.
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import load_reference as LR          # noqa: E402
from oracle import unet_ref as U                 # noqa: E402
from tests.golden.recipes import rnd, pack, pack_dict, digest   # noqa: E402

torch.set_num_threads(1)                          # single-thread: reproducible reductions
ref = LR.load()
TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1,
            apply_attn=[False, True], drop_rate=0.0)


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def randomized(module, seed):
    sd = U.randomize_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return sd


# ----------------------------------------------------------------------------- G1 op level
def g1_ops():
    torch.manual_seed(101)                        # module constructors draw their initial weights from the global RNG: seeded, so that
    out = {}                                      # re-running this script reproduces the committed files bit for bit
    # GroupNorm(32, eps 1e-6) (+SiLU): unet.py:18-20,85
    for B, C, hw in [(2, 128, 4), (1, 384, 16), (2, 768, 4), (1, 128, 16)]:
        gn = ref.unet.DEFAULT_NORMALIZER(C)
        sd = randomized(gn, 100 + C + hw)
        rec = dict(shape=(B, C, hw, hw), seed=C * 7 + hw, scale=1.7, shift=0.3)      # x = recipes.rnd(*shape, seed=.., ..)
        x = rnd(*rec["shape"], seed=rec["seed"], scale=rec["scale"], shift=rec["shift"])
        y = gn(x)
        out[f"gn_C{C}_hw{hw}"] = dict(x_recipe=rec, x_digest=digest(x), weight=sd["weight"], bias=sd["bias"], y=y.detach(),
                                      y_silu_digest=digest(F.silu(y)))
    # 3x3 s1 p1 convs incl. Cin=3 and Cout=3: modules.py:66-123
    for ci, co, hw in [(3, 32, 8), (32, 3, 8), (64, 32, 8)]:
        conv = ref.modules.Conv2d(ci, co, 3, 1, 1)
        sd = randomized(conv, 200 + ci)
        x = rnd(2, ci, hw, hw, seed=300 + ci)
        out[f"conv3_{ci}_{co}"] = dict(x=x, weight=sd["weight"], bias=sd["bias"], y=conv(x).detach())
    # SAME-pad stride-2 conv on even and odd H: modules.py:145-160, unet.py:165-167
    for hw in (8, 9):
        pad, conv = ref.modules.SamePad2d(3, 2), ref.modules.Conv2d(32, 32, 3, 2)
        sd = randomized(conv, 400 + hw)
        x = rnd(2, 32, hw, hw, seed=410 + hw)
        out[f"down_hw{hw}"] = dict(x=x, weight=sd["weight"], bias=sd["bias"], y=conv(pad(x)).detach())
    # 1x1 conv
    conv = ref.modules.Conv2d(64, 96, 1)
    sd = randomized(conv, 500)
    x = rnd(2, 64, 4, 4, seed=501)
    out["conv1x1"] = dict(x=x, weight=sd["weight"], bias=sd["bias"], y=conv(x).detach())
    # attention core: unet.py:43-51
    for C, h in [(64, 4), (256, 16)]:
        shp = (1 if C == 256 else 2, C, h, h)
        q, k, v = (rnd(*shp, seed=600 + i + C) for i in range(3))
        o = ref.unet.AttentionBlock.qkv(q, k, v)
        out[f"qkv_C{C}_L{h * h}"] = dict(shape=shp, seeds=[600 + i + C for i in range(3)], q_digest=digest(q), out=o)
    # nearest 2x upsample + conv: unet.py:199-202
    conv = ref.modules.Conv2d(32, 32, 3, 1, 1)
    sd = randomized(conv, 700)
    x = rnd(2, 32, 4, 4, seed=701)
    out["up_conv"] = dict(x=x, weight=sd["weight"], bias=sd["bias"],
                          y=conv(torch.nn.Upsample(scale_factor=2, mode="nearest")(x)).detach())
    # timestep embedding: functions.py:10-26
    t = torch.tensor([0, 1, 500, 999])
    for dim in (128, 127):
        out[f"temb_{dim}"] = dict(t=t, emb=ref.functions.get_timestep_embedding(t, dim))
    save("g1_ops.pt", out)


# ----------------------------------------------------------------------------- G2 block level
def _grads(module, loss):
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in module.named_parameters()}


def g2_blocks():
    torch.manual_seed(202)                        # (as in g1_ops)
    out = {}
    res = ref.unet.ResidualBlock(32, 64, embed_dim=128, drop_rate=0.0)
    sd = randomized(res, 11)
    x = rnd(2, 32, 8, 8, seed=12).requires_grad_(True)
    te = rnd(2, 128, seed=13).requires_grad_(True)
    gy = rnd(2, 64, 8, 8, seed=14)
    y = res(x, t_emb=te)
    g = _grads(res, (y * gy).sum())
    out["res"] = dict(sd=sd, x=x.detach(), t_emb=te.detach(), gy=gy, y=y.detach(), gx=x.grad, gt_emb=te.grad, grads=g)
    att = ref.unet.AttentionBlock(64)
    sd = randomized(att, 21)
    x = rnd(2, 64, 4, 4, seed=22).requires_grad_(True)
    gy = rnd(2, 64, 4, 4, seed=23)
    y = att(x)
    g = _grads(att, (y * gy).sum())
    out["attn"] = dict(sd=sd, x=x.detach(), gy=gy, y=y.detach(), gx=x.grad, grads=g)
    save("g2_blocks.pt", out)


# ----------------------------------------------------------------------------- G3 model level
def tiny_model(seed=1234):
    torch.manual_seed(seed)
    m = ref.UNet(**TINY)
    init = {k: v.clone() for k, v in m.state_dict().items()}
    sd = randomized(m, 31)
    return m, init, sd


def g3_model():
    out = {"tiny_cfg": TINY}
    m, init, sd = tiny_model()
    out["tiny_init_seed"], out["tiny_rand_seed"] = 1234, 31     # sd == randomize_state_dict(init(seed 1234), 31)
    out["tiny_init_sd"] = pack_dict(init)
    out["tiny_sd"] = pack_dict(sd)
    x = rnd(2, 3, 8, 8, seed=32)
    t = torch.tensor([7, 912])
    gy = rnd(2, 3, 8, 8, seed=33)
    m.train()
    y = m(x, t)
    (y * gy).sum().backward()
    full = ("in_conv.weight", "downsamples.level_0.0.conv1.weight", "upsamples.level_0.1.conv1.weight", "out_conv.2.weight",
            "middle.1.project_in.weight", "downsamples.level_0.1.1.weight", "upsamples.level_1.2.1.weight")
    out["tiny"] = dict(x=x, t=t, gy=gy, y=y.detach(),
                       grads=pack_dict({k: p.grad for k, p in m.named_parameters()}, full_keys=full))
    # the reference's own __main__ smoke config (unet.py:237): output checksum only
    torch.manual_seed(99)
    big = ref.UNet(3, 128, 3, (1, 2, 3), 2, (False, True, False))
    randomized(big, 41)
    big.eval()
    xs = rnd(2, 3, 32, 32, seed=42)
    ts = torch.tensor([0, 999])
    with torch.no_grad():
        ys = big(xs, ts)
    out["smoke"] = dict(init_seed=99, rand_seed=41, x_seed=42, t=ts, y_sum=ys.double().sum(), y_abs_sum=ys.double().abs().sum(),
                        y_corner=ys[:, :, :4, :4].clone())
    # state-dict key lists + shapes for the shipped configs
    for name in ("cifar10", "celeba", "celebahq"):
        cfg = json.load(open(os.path.join(LR.REFERENCE_ROOT, "configs", name + ".json")))
        mc = dict(cfg["model"]); mc.pop("block_size", None)
        mc["out_channels"] = mc["in_channels"]
        with torch.device("meta"):
            net = ref.UNet(**mc)
        out["keys_" + name] = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        out["cfg_" + name] = cfg
    # init parity for the CIFAR net: checksums of every tensor under seed 1234
    cfg = dict(out["cfg_cifar10"]["model"]); cfg["out_channels"] = 3
    torch.manual_seed(1234)
    net = ref.UNet(**cfg)
    out["cifar_init_sums"] = {k: float(v.double().sum()) for k, v in net.state_dict().items()}
    save("g3_model.pt", out)


# ----------------------------------------------------------------------------- G4 tables
def _tables(obj):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in obj.__dict__.items()}


def g4_tables():
    out = {}
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    for vt in ("fixed-small", "fixed-large"):
        out["ddpm_" + vt] = _tables(ref.GaussianDiffusion(betas, "eps", vt, "mse"))
    out["toy_fixed-large"] = _tables(ref.GaussianDiffusion(ref.get_beta_schedule("linear", 1e-3, 0.2, 100), "eps", "fixed-large", "mse"))
    for kind in ("quad", "warmup10", "warmup50", "const", "jsd"):
        out["betas_" + kind] = ref.get_beta_schedule(kind, 1e-4, 0.02, 1000)
    for sched, size in (("linear", 50), ("quadratic", 100), ("quadratic", 50)):
        sub = ref.get_selection_schedule(sched, size, 1000)
        out[f"sel_{sched}_{size}"] = sub
        if (sched, size) == ("quadratic", 50):
            continue
        for eta in (0.0, 1.0):
            for vt in ("fixed-small", "fixed-large"):
                out[f"ddim_{sched}_{size}_eta{eta}_{vt}"] = _tables(ref.DDIM(betas, "eps", vt, "mse", eta=eta, subsequence=sub))
    save("g4_tables.pt", out)


# ----------------------------------------------------------------------------- G5 step level
def g5_steps():
    out = {}
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    m, _, _ = tiny_model()
    m.eval()
    x0 = rnd(4, 3, 8, 8, seed=51).clamp(-1, 1)
    noise = rnd(4, 3, 8, 8, seed=52)
    z = rnd(4, 3, 8, 8, seed=53)
    t = torch.tensor([0, 1, 500, 999])
    lin = lambda x, t: 0.1 * x + 0.01 * t.reshape(-1, 1, 1, 1).to(x)        # closed-form denoiser
    for vt in ("fixed-small", "fixed-large"):
        dif = ref.GaussianDiffusion(betas, "eps", vt, "mse")
        x_t = dif.q_sample(x0, t, noise)
        rec = dict(x0=x0, noise=noise, z=z, t=t, x_t=x_t)
        for name, fn in (("lin", lin), ("unet", m)):
            with torch.no_grad():
                rec["loss_" + name] = dif.train_losses(fn, x0, t, noise=noise)
                mean, var, logvar, px0 = dif.p_mean_var(fn, x_t, t, clip_denoised=True, return_pred=True)
                rec["mean_" + name], rec["pred_x0_" + name] = mean, px0
                rec["var"], rec["logvar"] = var, logvar

                class G:    # inject z instead of the generator draw (diffusion.py:155)
                    pass
                orig = torch.Tensor.normal_
                torch.Tensor.normal_ = lambda self, *a, **k: self.copy_(z)
                try:
                    rec["x_prev_" + name] = dif.p_sample_step(fn, x_t, t)
                finally:
                    torch.Tensor.normal_ = orig
        out[vt] = rec
    save("g5_steps.pt", out)


# ----------------------------------------------------------------------------- G6 loop level
def g6_loops():
    out = {}
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    m, _, _ = tiny_model()
    m.eval()
    shape = (1, 3, 8, 8)
    for vt in ("fixed-large", "fixed-small"):
        dif = ref.GaussianDiffusion(betas, "eps", vt, "mse")
        x = dif.p_sample(m, shape=shape, device=torch.device("cpu"), seed=7)
        # the noise stream the reference consumed: x_T first, then one z per step (diffusion.py:164-173)
        g = torch.Generator("cpu").manual_seed(7)
        x_T = torch.empty(shape).normal_(generator=g)
        zs = torch.stack([torch.empty(shape).normal_(generator=g) for _ in range(1000)])
        out["ddpm_" + vt] = dict(seed=7, shape=shape, x_0=x, x_T=x_T, zs_sum=zs.double().sum(), zs_abs_sum=zs.double().abs().sum(),
                                 z_first=zs[0].clone(), z_last=zs[-1].clone())
        if vt == "fixed-large":
            # p_sample_progressive (diffusion.py:176-198): same stream, pred_x0 kept every 100 steps
            xp, preds = dif.p_sample_progressive(m, shape=shape, device=torch.device("cpu"), pred_freq=100, seed=7)
            assert torch.equal(xp, x)
            out["ddpm_" + vt].update(pred_freq=100, preds=preds)
    base = ref.GaussianDiffusion(betas, "eps", "fixed-small", "mse")
    for sched, size, eta in (("linear", 50, 0.0), ("quadratic", 100, 1.0)):
        sub = ref.get_selection_schedule(sched, size, 1000)
        ddim = ref.DDIM.from_ddpm(base, eta=eta, subsequence=sub)
        x = ddim.p_sample(m, shape=(2, 3, 8, 8), device=torch.device("cpu"), seed=11)
        out[f"ddim_{sched}_{size}_eta{eta}"] = dict(seed=11, shape=(2, 3, 8, 8), x_0=x)
    save("g6_loops.pt", out)


# ----------------------------------------------------------------------------- G7 train steps
def g7_train():
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    dif = ref.GaussianDiffusion(betas, "eps", "fixed-large", "mse")
    m, _, sd0 = tiny_model()
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: min((t + 1) / 5000, 1.0))
    tr = ref.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0,
                     shape=(3, 8, 8), device=torch.device("cpu"), ema_decay=0.9999)
    m.train()
    xs = [(torch.rand(4, 3, 8, 8, generator=torch.Generator().manual_seed(70 + i)) * 2 - 1) for i in range(3)]
    losses = []
    for i, x in enumerate(xs):
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    out = dict(cfg=TINY, init_seed=1234, rand_seed=31, sd0=pack_dict(sd0), xs=xs, losses=torch.tensor(losses, dtype=torch.float64), lr=2e-4, warmup=5000, gen_seed=8191,
               params=pack_dict(m.state_dict()), shadow=pack_dict(tr.ema.shadow), num_updates=tr.ema.num_updates,
               last_lr=sched.get_last_lr()[0])
    save("g7_train.pt", out)


# ----------------------------------------------------------------------------- G9 train steps that MOVE the weights
def g9_train_lr():
    """Like G7 but with a learning rate that changes the parameters by ~1e-2 per step and no warm-up: a forward that keeps
    using stale derived weight copies after an optimiser step cannot reproduce these losses (G7's effective lr of 4e-8 could)."""
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    dif = ref.GaussianDiffusion(betas, "eps", "fixed-large", "mse")
    m, _, sd0 = tiny_model()
    lr = 3e-3
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: 1.0 if t < 3 else 0.5)
    tr = ref.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0,
                     shape=(3, 8, 8), device=torch.device("cpu"), ema_decay=0.9999)
    m.train()
    xs = [(torch.rand(4, 3, 8, 8, generator=torch.Generator().manual_seed(90 + i)) * 2 - 1) for i in range(6)]
    losses = []
    for i, x in enumerate(xs):
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    out = dict(cfg=TINY, init_seed=1234, rand_seed=31, xs=xs, losses=torch.tensor(losses, dtype=torch.float64), lr=lr, gen_seed=8191,
               params=pack_dict(m.state_dict()), shadow=pack_dict(tr.ema.shadow), num_updates=tr.ema.num_updates,
               last_lr=sched.get_last_lr()[0])
    save("g9_train_lr.pt", out)


# ----------------------------------------------------------------------------- G8 toy plumbing
def g8_toy():
    torch.manual_seed(1234)
    dec = ref.toy_model.Decoder(2, 128, 3)
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    x = rnd(16, 2, seed=81)
    t = torch.arange(16) * 6
    gy = rnd(16, 2, seed=82)
    y = dec(x, t)
    (y * gy).sum().backward()
    grads = {k: p.grad.clone() for k, p in dec.named_parameters()}
    nparams = sum(p.numel() for p in dec.parameters())
    dif = ref.toy_diffusion.GaussianDiffusion(ref.get_beta_schedule("linear", 1e-3, 0.2, 100), "eps", "fixed-large", "mse")
    dec.zero_grad()
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(83)
    # gaussian8-like data: 8 modes on a circle (shape of the toy data, not the reference's streamer)
    ang = torch.randint(8, (10, 1000), generator=g).double() * (3.141592653589793 / 4)
    data = (torch.stack([ang.cos(), ang.sin()], -1) * 2 + 0.1 * torch.randn(10, 1000, 2, generator=g, dtype=torch.float64)).float()
    ts = torch.randint(100, (10, 1000), generator=g)
    noises = torch.randn(10, 1000, 2, generator=g)
    losses = []
    for i in range(10):
        loss = dif.train_losses(dec, data[i], ts[i], noise=noises[i]).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    out = dict(sd=sd, x=x, t=t, gy=gy, y=y.detach(), grads=pack_dict(grads), nparams=nparams,
               data_seed=83, losses=torch.tensor(losses, dtype=torch.float64),
               sd_after=pack_dict(dec.state_dict()))
    save("g8_toy.pt", out)



# ----------------------------------------------------------------------------- G10 the north-star sentence on CONFIG 2 ITSELF
def _shipped_model(name, init_seed, rand_seed):
    cfg = json.load(open(os.path.join(LR.REFERENCE_ROOT, "configs", name + ".json")))
    mc = dict(cfg["model"]); mc.pop("block_size", None)
    mc["out_channels"] = mc["in_channels"]
    torch.manual_seed(init_seed)
    m = ref.UNet(**mc)
    randomized(m, rand_seed)                      # the zero-initialised layers (conv2, project_out, out_conv) get weights: every path carries signal
    return m.eval(), mc


def g10_config2():
    """UNet forward and the sampling loops of the SHIPPED configurations (configs/cifar10.json at 32 x 32 — BASELINE config 2 — and
    configs/celeba.json at 64 x 64 for DDIM-50), where the product's hot kernels run (the G6 net is an 8 x 8 toy).  The 35.7 M weights
    do not travel: the fixture keeps the seeds (seeded init is bit-identical in the product, tests/test_unet_gpu.py::test_g3_keys_and_init,
    then oracle.unet_ref.randomize_state_dict), the inputs and the reference's outputs.  ~3 minutes per 1000-step chain on one core."""
    out = {}
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    m, mc = _shipped_model("cifar10", 1234, 61)
    out["cifar"] = dict(cfg=mc, init_seed=1234, rand_seed=61)
    x, t = rnd(2, 3, 32, 32, seed=62), torch.tensor([3, 977])
    with torch.no_grad():
        out["cifar"]["fwd"] = dict(x_seed=62, t=t, y=m(x, t))
    shape = (1, 3, 32, 32)
    dif = ref.GaussianDiffusion(betas, "eps", "fixed-large", "mse")
    xp, preds = dif.p_sample_progressive(m, shape=shape, device=torch.device("cpu"), pred_freq=250, seed=7)      # diffusion.py:176-198
    g = torch.Generator("cpu").manual_seed(7)
    x_T = torch.empty(shape).normal_(generator=g)
    zs = torch.stack([torch.empty(shape).normal_(generator=g) for _ in range(1000)])
    out["cifar"]["ddpm_fixed-large"] = dict(seed=7, shape=shape, x_0=xp, pred_freq=250, preds=preds, x_T_sum=x_T.double().sum(),
                                             zs_sum=zs.double().sum(), zs_abs_sum=zs.double().abs().sum())
    # p_sample (diffusion.py:160-174) consumes the same stream and must end on the same sample: 40 steps of it are enough to pin that
    short = ref.GaussianDiffusion(ref.get_beta_schedule("linear", 1e-4, 0.02, 40), "eps", "fixed-large", "mse")
    out["cifar"]["ddpm40_fixed-large"] = dict(seed=9, shape=(2, 3, 32, 32), timesteps=40,
                                               x_0=short.p_sample(m, shape=(2, 3, 32, 32), device=torch.device("cpu"), seed=9))
    m2, mc2 = _shipped_model("celeba", 4321, 63)
    base = ref.GaussianDiffusion(betas, "eps", "fixed-small", "mse")
    ddim = ref.DDIM.from_ddpm(base, eta=0.0, subsequence=ref.get_selection_schedule("linear", 50, 1000))        # ddim.py:96-113
    out["celeba"] = dict(cfg=mc2, init_seed=4321, rand_seed=63,
                         ddim_linear_50=dict(seed=11, shape=(1, 3, 64, 64), x_0=ddim.p_sample(m2, shape=(1, 3, 64, 64), device=torch.device("cpu"), seed=11)))
    save("g10_config2.pt", out)


def strided(t, n=256):
    """n evenly spaced elements of a tensor (its flat order): a sample that reaches every region of a tensor too large to commit."""
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).round().long()
    return f[idx].clone()


def g11_config2_bench_batch():
    """BASELINE config 2 AT THE BENCHMARK'S BATCH (B = 128): the geometry at which the product dispatches its hot kernels — the
    wave-specialised / persistent 3x3 kernels need >= 16384 / 4096 pixels per layer, the streaming 1x1 kernel >= 8192, the patch-stationary
    weight gradient whole 16 x 16 patches — which the B = 1 / B = 2 records of G10 never reach.  From the imported reference
    (models/unet.py:205-233, diffusion.py:160-174): an eval forward, EVERY parameter gradient of sum(y * gy) (dropout inactive: eval mode,
    the product side builds the net with drop_rate 0), and an 8-step ancestral chain.  Tensors travel as strided samples + fp64 digests;
    inputs are seeds.  ~15 minutes on one core."""
    B = 128
    m, mc = _shipped_model("cifar10", 1234, 61)
    out = dict(cfg=mc, init_seed=1234, rand_seed=61, B=B)
    x, gy = rnd(B, 3, 32, 32, seed=71), rnd(B, 3, 32, 32, seed=72)
    t = (torch.arange(B) * 37 + 5) % 1000
    for q in m.parameters():
        q.requires_grad_(True)
    y = m(x, t)
    (y * gy).sum().backward()
    yd = y.detach()
    out["fwd"] = dict(x_seed=71, t=t, y_sub=yd[:, :, ::4, ::4].clone(), y_sum=yd.double().sum((1, 2, 3)), y_abs=yd.double().abs().sum((1, 2, 3)),
                      y_absmax=float(yd.abs().max()))
    names, sums, abss, sqs, samples = [], [], [], [], {}
    for k, q in m.named_parameters():
        g = q.grad.detach()
        names.append(k); sums.append(float(g.double().sum())); abss.append(float(g.double().abs().sum())); sqs.append(float((g.double() ** 2).sum()))
        samples[k] = strided(g)
    out["grads"] = dict(gy_seed=72, names=names, sum=torch.tensor(sums, dtype=torch.float64), abs_sum=torch.tensor(abss, dtype=torch.float64),
                        sq_sum=torch.tensor(sqs, dtype=torch.float64), samples=samples)
    for q in m.parameters():
        q.requires_grad_(False); q.grad = None
    steps, shape = 8, (B, 3, 32, 32)
    short = ref.GaussianDiffusion(ref.get_beta_schedule("linear", 1e-4, 0.02, steps), "eps", "fixed-large", "mse")
    x0 = short.p_sample(m, shape=shape, device=torch.device("cpu"), seed=73)                 # diffusion.py:160-174
    out["ddpm8_fixed-large"] = dict(seed=73, shape=shape, timesteps=steps, x0_sub=x0[:, :, ::4, ::4].clone(), x0_sum=x0.double().sum((1, 2, 3)),
                                    x0_abs=x0.double().abs().sum((1, 2, 3)), x0_absmax=float(x0.abs().max()))
    save("g11_config2_b128.pt", out)


def g12_config2_train_steps():
    """Three `Trainer.step`s (utils/train.py:148-170: q_sample, loss, backward, clip_grad_norm_(1.0), Adam, EMA :300-305) of the
    configs/cifar10.json network AT THE BENCHMARK'S BATCH (B = 128, 32 x 32), by the reference's own Trainer on its CPU (t, noise) stream.
    Dropout is 0 here (the reference draws its masks from torch's CPU generator, which no other implementation can reproduce) and the
    learning rate is 1e-3 without warm-up, so that the second and third loss depend on the weights the first steps wrote.  Inputs are
    seeds; parameters and EMA shadows travel as fp64 sums + 64 strided entries per tensor.  ~8 minutes on one core."""
    B = 128
    cfg = json.load(open(os.path.join(LR.REFERENCE_ROOT, "configs", "cifar10.json")))
    mc = dict(cfg["model"]); mc.pop("block_size", None)
    mc["out_channels"] = mc["in_channels"]; mc["drop_rate"] = 0.0
    torch.manual_seed(1234)
    m = ref.UNet(**mc)
    randomized(m, 61)
    dif = ref.GaussianDiffusion(ref.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    lr = 1e-3
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: 1.0)
    tr = ref.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0,
                     shape=(3, 32, 32), device=torch.device("cpu"), ema_decay=0.9999)
    m.train()
    x_seeds = [121, 122, 123]
    losses = []
    for i, sd in enumerate(x_seeds):
        x = torch.rand(B, 3, 32, 32, generator=torch.Generator().manual_seed(sd)) * 2 - 1
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
        print("step", i + 1, losses[-1], flush=True)

    def digest_state(state):
        names = list(state)
        return dict(names=names, sum=torch.tensor([float(state[k].double().sum()) for k in names], dtype=torch.float64),
                    abs_sum=torch.tensor([float(state[k].double().abs().sum()) for k in names], dtype=torch.float64),
                    samples={k: strided(state[k], 64) for k in names})
    params = {k: v.detach() for k, v in m.named_parameters()}
    out = dict(cfg=mc, init_seed=1234, rand_seed=61, B=B, lr=lr, gen_seed=8191, x_seeds=x_seeds, losses=torch.tensor(losses, dtype=torch.float64),
               params=digest_state(params), shadow=digest_state({k: tr.ema.shadow[k] for k in params}), num_updates=tr.ema.num_updates)
    save("g12_config2_train_b128.pt", out)


def g13_config5_and_4_at_their_batches():
    """BASELINE config 5's per-GPU work — configs/celebahq.json (113.7 M parameters, six levels, 512-channel attention at 16 x 16) at
    256 x 256, B = 2 — forward and EVERY parameter gradient of sum(y * gy) from the imported reference (dropout is 0 in this config), and
    config 4's network (configs/celeba.json) at 64 x 64 and B = 32, where its layers reach the large-grid kernels: eval forward.
    Strided samples + fp64 sums, inputs as seeds.  ~5 minutes on one core."""
    out = {}
    m, mc = _shipped_model("celebahq", 2345, 64)
    m.train()
    B = 2
    x, gy, t = rnd(B, 3, 256, 256, seed=131), rnd(B, 3, 256, 256, seed=132), torch.tensor([417, 36])
    for q in m.parameters():
        q.requires_grad_(True)
    y = m(x, t)
    (y * gy).sum().backward()
    yd = y.detach()
    names = [k for k, _ in m.named_parameters()]
    grads = {k: q.grad.detach() for k, q in m.named_parameters()}
    out["celebahq"] = dict(cfg=mc, init_seed=2345, rand_seed=64, B=B, x_seed=131, gy_seed=132, t=t,
                           y_sub=yd[:, :, ::16, ::16].clone(), y_sum=yd.double().sum((1, 2, 3)), y_abs=yd.double().abs().sum((1, 2, 3)), y_absmax=float(yd.abs().max()),
                           grads=dict(names=names, sum=torch.tensor([float(grads[k].double().sum()) for k in names], dtype=torch.float64),
                                      abs_sum=torch.tensor([float(grads[k].double().abs().sum()) for k in names], dtype=torch.float64),
                                      sq_sum=torch.tensor([float((grads[k].double() ** 2).sum()) for k in names], dtype=torch.float64),
                                      samples={k: strided(grads[k], 64) for k in names}))
    del m, grads, y
    m2, mc2 = _shipped_model("celeba", 4321, 63)
    B2 = 32
    x2, t2 = rnd(B2, 3, 64, 64, seed=133), (torch.arange(B2) * 31 + 7) % 1000
    with torch.no_grad():
        y2 = m2(x2, t2)
    out["celeba"] = dict(cfg=mc2, init_seed=4321, rand_seed=63, B=B2, x_seed=133, t=t2, y_sub=y2[:, :, ::8, ::8].clone(),
                         y_sum=y2.double().sum((1, 2, 3)), y_abs=y2.double().abs().sum((1, 2, 3)), y_absmax=float(y2.abs().max()))
    save("g13_config5_config4.pt", out)


def g14_config4_ddim50_at_its_batch():
    """BASELINE config 4 AT ITS BATCH: configs/celeba.json at 64 x 64, the 50-step eta = 0 DDIM chain (ddim.py:96-113 on top of
    diffusion.py:160-174) for B = 128 samples at once — the geometry at which the product serves the chain from its large-grid kernels
    (G10 pins the same chain at B = 1, where the small-grid kernels run).  From the imported reference on its CPU noise stream; the result
    travels as an 8 x 8-strided sample of every image + per-image fp64 sums.  50 x 128 forwards of 46.7 GFLOP: ~45 minutes on 6 cores."""
    B = 128
    torch.set_num_threads(int(os.environ.get("G14_THREADS", "6")))      # (the other fixtures are written single-threaded; this one is 300 TFLOP)
    m2, mc2 = _shipped_model("celeba", 4321, 63)
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    base = ref.GaussianDiffusion(betas, "eps", "fixed-small", "mse")
    ddim = ref.DDIM.from_ddpm(base, eta=0.0, subsequence=ref.get_selection_schedule("linear", 50, 1000))
    shape = (B, 3, 64, 64)
    with torch.no_grad():
        x0 = ddim.p_sample(m2, shape=shape, device=torch.device("cpu"), seed=141)
    out = dict(cfg=mc2, init_seed=4321, rand_seed=63, B=B, seed=141, shape=shape, steps=50, x0_sub=x0[:, :, ::8, ::8].clone(),
               x0_sum=x0.double().sum((1, 2, 3)), x0_abs=x0.double().abs().sum((1, 2, 3)), x0_absmax=float(x0.abs().max()),
               x0_first=x0[0].clone())
    save("g14_config4_ddim50_b128.pt", out)


def g15_metrics():
    """The evaluation arithmetic from the reference's own functions on synthetic features:
    * `InceptionStatistics.forward` / `get_statistics` (metrics/fid_score.py:109-139) fed through an identity "network" in three unequal
      batches; `calculate_frechet_distance` (:264-316) between two such statistics;
    * `ManifoldBuilder.compute_kth` (metrics/precision_recall.py:159-168) and `calc_pr` (:177-206) on two feature clouds.
    The feature networks themselves (pretrained Inception-v3 / VGG-16) cannot be fetched here; everything around them is pinned."""
    import numpy as np
    rm = LR.load_metrics()
    from tests.golden.recipes import g15_inputs
    D = 48
    gen, real = g15_inputs(D, 151)

    def stats(x, sizes):
        st = object.__new__(rm.fid_score.InceptionStatistics)          # (its __init__ loads the pretrained network)
        st.input_transform = lambda v: v
        st.activation_dim = D
        st.model = lambda v: v
        st.running_mean = np.zeros((D,), dtype=np.float64)
        st.running_var = np.zeros((D, D), dtype=np.float64)
        st.count = 0
        o = 0
        for n in sizes:
            st.forward(x[o:o + n].reshape(n, D, 1, 1))
            o += n
        assert o == x.shape[0]
        return st.get_statistics()
    mu_g, cov_g = stats(gen, (256, 256, 188))
    mu_r, cov_r = stats(real, (100, 500, 300))
    fd = rm.fid_score.calc_fd(mu_g, cov_g, mu_r, cov_r)
    fd_self = rm.fid_score.calc_fd(mu_g, cov_g, mu_g, cov_g)
    pr = rm.precision_recall
    fa, fb = gen[:400].contiguous(), real[:500].contiguous() * 0.9

    def kth_of(f, k):
        b = object.__new__(pr.ManifoldBuilder)                          # (its __init__ casts to fp16, which torch.cdist lacks on the host)
        b.nhood_size, b.row_batch_size, b.col_batch_size, b.op_device = k, 128, 200, torch.device("cpu")
        return b.compute_kth(f)
    ka, kb = kth_of(fa, 3), kth_of(fb, 3)
    precision, recall = pr.calc_pr(pr.Manifold(fa, ka), pr.Manifold(fb, kb), row_batch_size=128, col_batch_size=200, device=torch.device("cpu"))
    k5 = kth_of(fa, 5)
    out = dict(D=D, seed=151, gen_sum=gen.double().sum(), real_sum=real.double().sum(), gen_batches=(256, 256, 188), real_batches=(100, 500, 300),
               mu_g=torch.from_numpy(mu_g), cov_g=torch.from_numpy(cov_g), mu_r=torch.from_numpy(mu_r), cov_r=torch.from_numpy(cov_r),
               fd=float(fd), fd_self=float(fd_self), kth_a=ka, kth_b=kb, kth_a5=k5,
               precision=float(precision), recall=float(recall),
               to_uint8_in=torch.linspace(-1.2, 1.2, 41), to_uint8_out=pr.to_uint8(torch.linspace(-1.2, 1.2, 41)))
    print("G15: fd", fd, "fd_self", fd_self, "precision", float(precision), "recall", float(recall))
    save("g15_metrics.pt", out)


if __name__ == "__main__":
    import sys
    ALL = dict(g1=g1_ops, g2=g2_blocks, g3=g3_model, g4=g4_tables, g5=g5_steps, g6=g6_loops, g7=g7_train, g8=g8_toy, g9=g9_train_lr, g10=g10_config2,
               g11=g11_config2_bench_batch, g12=g12_config2_train_steps, g13=g13_config5_and_4_at_their_batches,
               g14=g14_config4_ddim50_at_its_batch, g15=g15_metrics)
    for name in (sys.argv[1:] or list(ALL)):          # `make_golden.py g9` regenerates one fixture, no argument = all
        ALL[name]()
