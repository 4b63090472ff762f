"""Pin the oracle (oracle/*.py) against the fixtures generated from the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances: the oracle runs the same ATen CPU
kernels as the reference, so fp32 results agree to ~1e-6 relative; tables are fp64-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import diffusion_ref as D
from oracle import toy_ref
from oracle import train_ref
from oracle import unet_ref as U
from tests.golden.recipes import check, rnd

RT = 2e-5


def test_g1_groupnorm(golden):
    g = golden("g1_ops.pt")
    for k, rec in g.items():
        if not k.startswith("gn_"):
            continue
        r = rec["x_recipe"]
        x = rnd(*r["shape"], seed=r["seed"], scale=r["scale"], shift=r["shift"])
        check(x, rec["x_digest"], 1e-6, name=k + ".x")
        y = U.group_norm(x, rec["weight"], rec["bias"])
        check(y, rec["y"], RT, name=k)
        check(F.silu(y), rec["y_silu_digest"], RT, name=k + ".silu")


def test_g1_convs(golden):
    g = golden("g1_ops.pt")
    for k in ("conv3_3_32", "conv3_32_3", "conv3_64_32"):
        r = g[k]
        check(F.conv2d(r["x"], r["weight"], r["bias"], padding=1), r["y"], RT, name=k)
    for k in ("down_hw8", "down_hw9"):
        r = g[k]
        y = F.conv2d(U.same_pad_s2(r["x"]), r["weight"], r["bias"], stride=2)
        assert y.shape == r["y"].shape
        check(y, r["y"], RT, name=k)
    r = g["conv1x1"]
    check(F.conv2d(r["x"], r["weight"], r["bias"]), r["y"], RT, name="conv1x1")
    r = g["up_conv"]
    check(F.conv2d(F.interpolate(r["x"], scale_factor=2, mode="nearest"), r["weight"], r["bias"], padding=1), r["y"], RT, name="up_conv")


def test_g1_attention_and_temb(golden):
    g = golden("g1_ops.pt")
    for k in ("qkv_C64_L16", "qkv_C256_L256"):
        r = g[k]
        q, kk, v = (rnd(*r["shape"], seed=s) for s in r["seeds"])
        check(q, r["q_digest"], 1e-6, name=k + ".q")
        check(U.attention_core(q, kk, v), r["out"], RT, name=k)
    for dim in (128, 127):
        r = g[f"temb_{dim}"]
        check(U.timestep_embedding(r["t"], dim), r["emb"], 1e-6, name=f"temb{dim}")


def _leaf(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def test_g2_blocks(golden):
    g = golden("g2_blocks.pt")
    r = g["res"]
    sd = _leaf(r["sd"])
    x = r["x"].clone().requires_grad_(True)
    te = r["t_emb"].clone().requires_grad_(True)
    y = U.residual_block(sd, "", x, te)
    check(y, r["y"], RT, name="res.y")
    (y * r["gy"]).sum().backward()
    check(x.grad, r["gx"], RT, name="res.gx")
    check(te.grad, r["gt_emb"], RT, name="res.gt")
    for k, v in r["grads"].items():
        check(sd[k].grad, v, RT, name="res." + k)
    r = g["attn"]
    sd = _leaf(r["sd"])
    x = r["x"].clone().requires_grad_(True)
    y = U.attention_block(sd, "", x)
    check(y, r["y"], RT, name="attn.y")
    (y * r["gy"]).sum().backward()
    check(x.grad, r["gx"], RT, name="attn.gx")
    for k, v in r["grads"].items():
        check(sd[k].grad, v, RT, name="attn." + k)


def tiny_sd(g3):
    torch.manual_seed(g3["tiny_init_seed"])
    init = U.init_state_dict(g3["tiny_cfg"])
    return init, U.randomize_state_dict(init, g3["tiny_rand_seed"])


def test_g3_tiny_unet(golden):
    g = golden("g3_model.pt")
    init, sd = tiny_sd(g)
    for k, v in g["tiny_init_sd"].items():
        check(init[k], v, 0.0, atol=0.0, name="init." + k)          # bit-identical initialisation
    for k, v in g["tiny_sd"].items():
        check(sd[k], v, 1e-7, name="sd." + k)
    r = g["tiny"]
    p = _leaf(sd)
    y = U.unet_forward(p, g["tiny_cfg"], r["x"], r["t"], training=True)
    check(y, r["y"], RT, name="tiny.y")
    (y * r["gy"]).sum().backward()
    assert list(r["grads"].keys()) == list(p.keys())
    for k, v in r["grads"].items():
        check(p[k].grad, v, 5e-5, atol=2e-5, name="grad." + k)   # atol: grads that are analytically 0 (1 ch/group)


def test_g3_keys_and_init(golden):
    g = golden("g3_model.pt")
    for name in ("cifar10", "celeba", "celebahq"):
        cfg = dict(g["cfg_" + name]["model"]); cfg.pop("block_size", None)
        spec = [(k, tuple(s)) for k, s, _ in U.param_spec(cfg)]
        assert spec == [(k, tuple(s)) for k, s in g["keys_" + name]], name
    cfg = dict(g["cfg_cifar10"]["model"])
    torch.manual_seed(1234)
    sd = U.init_state_dict(cfg)
    for k, s in g["cifar_init_sums"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-12 * max(1.0, abs(s)), k   # fp64 sum order varies with threads
    assert sum(v.numel() for v in sd.values()) == 35746307
    assert sum(1 for v in sd.values() if v.ndim >= 2 and float(v.abs().max()) < 1e-5) == 29


def test_g3_smoke_config(golden):
    r = golden("g3_model.pt")["smoke"]
    cfg = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=(1, 2, 3), num_res_blocks=2,
               apply_attn=(False, True, False))
    torch.manual_seed(r["init_seed"])
    sd = U.randomize_state_dict(U.init_state_dict(cfg), r["rand_seed"])
    x = rnd(2, 3, 32, 32, seed=r["x_seed"])
    with torch.no_grad():
        y = U.unet_forward(sd, cfg, x, r["t"])
    check(y[:, :, :4, :4], r["y_corner"], RT, name="smoke.corner")
    assert abs(float(y.double().abs().sum()) - float(r["y_abs_sum"])) <= 1e-4 * float(r["y_abs_sum"])


def _cmp_tables(mine, gold, keys):
    for k in keys:
        a, b = mine[k], gold[k]
        assert a.dtype == torch.float64 or a.dtype == b.dtype
        assert torch.allclose(a, b, rtol=1e-13, atol=0), k


TABLE_KEYS = ["betas", "alphas_bar", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar", "sqrt_recip_alphas_bar",
              "sqrt_recip_m1_alphas_bar", "posterior_var", "posterior_logvar_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2", "fixed_model_var", "fixed_model_logvar"]


def test_g4_tables(golden):
    g = golden("g4_tables.pt")
    betas = D.beta_schedule("linear", 1e-4, 0.02, 1000)
    for vt in ("fixed-small", "fixed-large"):
        _cmp_tables(D.ddpm_tables(betas, vt), g["ddpm_" + vt], TABLE_KEYS)
    _cmp_tables(D.ddpm_tables(D.beta_schedule("linear", 1e-3, 0.2, 100), "fixed-large"), g["toy_fixed-large"], TABLE_KEYS)
    for kind in ("quad", "warmup10", "warmup50", "const", "jsd"):
        assert torch.equal(D.beta_schedule(kind, 1e-4, 0.02, 1000), g["betas_" + kind]), kind
    for sched, size in (("linear", 50), ("quadratic", 100), ("quadratic", 50)):
        assert torch.equal(D.selection_schedule(sched, size, 1000), g[f"sel_{sched}_{size}"])
    assert int((g["sel_quadratic_50"] == 0).sum()) == 2                    # documented duplicate
    for sched, size in (("linear", 50), ("quadratic", 100)):
        sub = D.selection_schedule(sched, size, 1000)
        for eta in (0.0, 1.0):
            for vt in ("fixed-small", "fixed-large"):
                gold = g[f"ddim_{sched}_{size}_eta{eta}_{vt}"]
                mine = D.ddim_tables(betas, vt, eta, sub)
                assert mine["model_var_type"] == gold["model_var_type"]
                _cmp_tables(mine, gold, TABLE_KEYS + ["alphas", "alphas_bar_prev", "sqrt_alphas_bar_prev"])
                assert torch.equal(mine["subsequence"], gold["subsequence"])
    # known answers
    T = D.ddpm_tables(betas, "fixed-large")
    assert float(T["posterior_var"][0]) == 0.0
    assert abs(float(T["fixed_model_logvar"][0]) - (-9.8167)) < 1e-3
    assert float(T["fixed_model_logvar"][0]) == float(torch.log(T["posterior_var"][1]))
    Td = D.ddim_tables(betas, "fixed-small", 0.0, D.selection_schedule("linear", 50, 1000))
    assert abs(float(Td["fixed_model_logvar"][5]) - math.log(1e-20)) < 1e-12


def test_g5_steps(golden):
    g = golden("g5_steps.pt")
    g3 = golden("g3_model.pt")
    _, sd = tiny_sd(g3)
    betas = D.beta_schedule("linear", 1e-4, 0.02, 1000)
    fns = dict(lin=lambda x, t: 0.1 * x + 0.01 * t.reshape(-1, 1, 1, 1).to(x),
               unet=lambda x, t: U.unet_forward(sd, g3["tiny_cfg"], x, t))
    for vt in ("fixed-small", "fixed-large"):
        r = g[vt]
        T = D.ddpm_tables(betas, vt)
        x_t = D.q_sample(T, r["x0"], r["t"], r["noise"])
        check(x_t, r["x_t"], 1e-6, name="x_t")
        for name, fn in fns.items():
            with torch.no_grad():
                eps = fn(r["x_t"], r["t"])
                check(D.mse_eps_loss(eps, r["noise"]), r["loss_" + name], RT, name="loss_" + name)
                xp, px0 = D.p_step_from_eps(T, r["x_t"], r["t"], eps, r["z"])
                check(px0, r["pred_x0_" + name], RT, name="px0_" + name)
                check(xp, r["x_prev_" + name], RT, name="xprev_" + name)
                # t = 0 row returns exactly the model mean (noise masked)
                assert torch.equal(xp[0], (r["mean_" + name])[0]) or float((xp[0] - r["mean_" + name][0]).abs().max()) < 1e-6


def _noise_stream(seed, shape, steps):
    g = torch.Generator("cpu").manual_seed(seed)
    x_T = torch.empty(shape).normal_(generator=g)
    zs = [torch.empty(shape).normal_(generator=g) for _ in range(steps)]
    return x_T, zs


def test_g6_loops(golden):
    g = golden("g6_loops.pt")
    g3 = golden("g3_model.pt")
    _, sd = tiny_sd(g3)
    fn = lambda x, t: U.unet_forward(sd, g3["tiny_cfg"], x, t)
    betas = D.beta_schedule("linear", 1e-4, 0.02, 1000)
    with torch.no_grad():
        for vt in ("fixed-large", "fixed-small"):
            r = g["ddpm_" + vt]
            x_T, zs = _noise_stream(r["seed"], tuple(r["shape"]), 1000)
            assert torch.equal(x_T, r["x_T"]) and torch.equal(zs[0], r["z_first"]) and torch.equal(zs[-1], r["z_last"])
            x = D.sample_loop(D.ddpm_tables(betas, vt), fn, x_T, zs)
            check(x, r["x_0"], 2e-4, name="ddpm_" + vt)
        for sched, size, eta in (("linear", 50, 0.0), ("quadratic", 100, 1.0)):
            r = g[f"ddim_{sched}_{size}_eta{eta}"]
            sub = D.selection_schedule(sched, size, 1000)
            T = D.ddim_tables(betas, "fixed-small", eta, sub)
            x_T, zs = _noise_stream(r["seed"], tuple(r["shape"]), size)
            x = D.sample_loop(T, fn, x_T, zs, timestep_map=sub)
            check(x, r["x_0"], 2e-4, name="ddim_" + sched)


def test_g10_config2_forward_and_short_chain(golden):
    """The oracle on BASELINE config 2 itself (configs/cifar10.json at 32 x 32): forward at B = 2 and the 40-step ancestral chain at
    B = 2 against the reference's outputs (the 1000-step chain of the fixture is left to the GPU side: minutes on one core here)."""
    g = golden("g10_config2.pt")["cifar"]
    torch.manual_seed(g["init_seed"])
    sd = U.randomize_state_dict(U.init_state_dict(g["cfg"]), g["rand_seed"])
    fn = lambda x, t: U.unet_forward(sd, g["cfg"], x, t)
    with torch.no_grad():
        f = g["fwd"]
        check(fn(rnd(2, 3, 32, 32, seed=f["x_seed"]), f["t"]), f["y"], 2e-5, name="g10.fwd")
        r = g["ddpm40_fixed-large"]
        x_T, zs = _noise_stream(r["seed"], tuple(r["shape"]), r["timesteps"])
        x = D.sample_loop(D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, r["timesteps"]), "fixed-large"), fn, x_T, zs)
        check(x, r["x_0"], 2e-4, name="g10.ddpm40")


def test_g14_oracle_ddim50_chain_of_config4_first_samples(golden):
    """The oracle on BASELINE config 4's chain (configs/celeba.json at 64 x 64, DDIM-50, eta = 0) against fixture G14, which the reference
    wrote at B = 128: eval-mode samples are independent, so the first TWO samples of the batch — x_T drawn for the whole batch, as the
    reference does — must come out the same (all 128 are compared on the GPU side: tests/test_config5_config4_gpu.py)."""
    g = golden("g14_config4_ddim50_b128.pt")
    torch.manual_seed(g["init_seed"])
    sd = U.randomize_state_dict(U.init_state_dict(g["cfg"]), g["rand_seed"])
    fn = lambda x, t: U.unet_forward(sd, g["cfg"], x, t)
    shape, n = tuple(g["shape"]), 2
    gen = torch.Generator("cpu").manual_seed(g["seed"])
    x_T = torch.empty(shape).normal_(generator=gen)[:n].clone()
    sub = D.selection_schedule("linear", g["steps"], 1000)
    T = D.ddim_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-small", 0.0, sub)
    with torch.no_grad():
        x = D.sample_loop(T, fn, x_T, [torch.zeros_like(x_T)] * g["steps"], timestep_map=sub)     # (eta = 0: the per-step draws are multiplied by zero)
    assert float((x[0] - g["x0_first"]).abs().max()) <= 2e-4 * g["x0_absmax"]
    assert float((x[:, :, ::8, ::8] - g["x0_sub"][:n]).abs().max()) <= 2e-4 * g["x0_absmax"]
    assert float(((x.double().sum((1, 2, 3)) - g["x0_sum"][:n]).abs() / g["x0_abs"][:n]).max()) < 1e-5


def test_g11_oracle_forward_and_gradients_at_the_bench_batch(golden):
    """The oracle at the benchmark's batch (configs/cifar10.json, B = 128, eval mode) against the reference's forward and a sample of its
    parameter gradients (fixture G11; the GPU side checks every gradient tensor: tests/test_config2_bench_batch_gpu.py)."""
    g = golden("g11_config2_b128.pt")
    torch.manual_seed(g["init_seed"])
    sd = U.randomize_state_dict(U.init_state_dict(g["cfg"]), g["rand_seed"])
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    f, gr = g["fwd"], g["grads"]
    B = g["B"]
    y = U.unet_forward(p, g["cfg"], rnd(B, 3, 32, 32, seed=f["x_seed"]), f["t"], training=False)
    (y * rnd(B, 3, 32, 32, seed=gr["gy_seed"])).sum().backward()
    yd = y.detach()
    assert float((yd[:, :, ::4, ::4] - f["y_sub"]).abs().max()) <= 2e-5 * f["y_absmax"]
    assert float(((yd.double().sum((1, 2, 3)) - f["y_sum"]).abs() / f["y_abs"]).max()) < 1e-6
    assert list(p) == gr["names"]
    for i, k in enumerate(gr["names"]):
        got = p[k].grad.reshape(-1)
        idx = torch.linspace(0, got.numel() - 1, min(256, got.numel())).round().long()
        want = gr["samples"][k]
        assert float((got[idx] - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1e-3), k
        assert abs(float(got.double().sum()) - float(gr["sum"][i])) <= 1e-5 * float(gr["abs_sum"][i]) + 1e-9, k


def test_g12_oracle_train_steps_at_the_bench_batch(golden):
    """oracle/train_ref.py against the reference's own Trainer on configs/cifar10.json at B = 128 (fixture G12: three steps at lr 1e-3,
    dropout 0, the reference's CPU (t, noise) stream): the losses and the sampled parameters / EMA shadows after the third step."""
    g = golden("g12_config2_train_b128.pt")
    torch.manual_seed(g["init_seed"])
    sd0 = U.randomize_state_dict(U.init_state_dict(g["cfg"]), g["rand_seed"])
    st = train_ref.TrainState(sd0, g["cfg"], lr=g["lr"], warmup=1, grad_norm=1.0, ema_decay=0.9999)
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-large")
    gen = torch.Generator("cpu").manual_seed(g["gen_seed"])
    losses = []
    for sd in g["x_seeds"]:
        x = torch.rand(g["B"], 3, 32, 32, generator=torch.Generator().manual_seed(sd)) * 2 - 1
        t = torch.empty((g["B"],), dtype=torch.int64).random_(to=1000, generator=gen)
        noise = torch.empty_like(x).normal_(generator=gen)
        losses.append(st.step(T, x, t, noise))
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=2e-4), (losses, g["losses"])
    assert st.num_updates == g["num_updates"]
    for state, dg in ((st.params, g["params"]), (st.shadow, g["shadow"])):
        beyond = total = 0
        for i, k in enumerate(dg["names"]):
            f = state[k].detach().reshape(-1)
            idx = torch.linspace(0, f.numel() - 1, min(64, f.numel())).round().long()
            d = (f[idx] - dg["samples"][k]).abs()
            beyond += int((d > 1e-3 * max(float(dg["samples"][k].abs().max()), 1e-3)).sum()); total += d.numel()
            assert float(d.max()) <= 2.0 * g["lr"] * len(g["x_seeds"]) + 1e-6, k          # inside Adam's reach
            assert abs(float(f.double().sum()) - float(dg["sum"][i])) <= 2e-3 * float(dg["abs_sum"][i]) + 1e-9, k
        assert beyond <= 0.02 * total, (beyond, total)


def test_g7_train_steps(golden):
    g = golden("g7_train.pt")
    torch.manual_seed(g["init_seed"])
    sd0 = U.randomize_state_dict(U.init_state_dict(g["cfg"]), g["rand_seed"])
    for k, v in g["sd0"].items():
        check(sd0[k], v, 1e-7, name="sd0." + k)
    st = train_ref.TrainState(sd0, g["cfg"], lr=g["lr"], warmup=g["warmup"], grad_norm=1.0, ema_decay=0.9999)
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-large")
    gen = torch.Generator("cpu").manual_seed(g["gen_seed"])                  # utils/train.py:115
    losses = []
    for x in g["xs"]:
        t = torch.empty((x.shape[0],), dtype=torch.int64).random_(to=1000, generator=gen)   # utils/train.py:138
        noise = torch.empty_like(x).normal_(generator=gen)                                   # utils/train.py:140
        losses.append(st.step(T, x, t, noise))
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=1e-5)
    assert st.num_updates == g["num_updates"] == 2
    assert abs(st.sched.get_last_lr()[0] - g["last_lr"]) < 1e-15
    for k, v in g["params"].items():
        check(st.params[k], v, 1e-5, name="param." + k)
    for k, v in g["shadow"].items():
        check(st.shadow[k], v, 1e-5, name="shadow." + k)
    # EMA decay schedule known answer: 0.1, 0.1818.., 0.25
    assert [round(min(0.9999, (1 + n) / (10 + n)), 4) for n in range(3)] == [0.1, 0.1818, 0.25]


def test_g9_train_steps_with_a_learning_rate_that_moves_the_weights(golden):
    g = golden("g9_train_lr.pt")
    torch.manual_seed(g["init_seed"])
    sd0 = U.randomize_state_dict(U.init_state_dict(g["cfg"]), g["rand_seed"])
    st = train_ref.TrainState(sd0, g["cfg"], lr=g["lr"], warmup=0, grad_norm=1.0, ema_decay=0.9999, lr_lambda=lambda s: 1.0 if s < 3 else 0.5)
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-large")
    gen = torch.Generator("cpu").manual_seed(g["gen_seed"])
    losses = []
    for x in g["xs"]:
        t = torch.empty((x.shape[0],), dtype=torch.int64).random_(to=1000, generator=gen)
        noise = torch.empty_like(x).normal_(generator=gen)
        losses.append(st.step(T, x, t, noise))
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=2e-4)
    assert abs(st.sched.get_last_lr()[0] - g["last_lr"]) < 1e-15
    moved = max(float((st.params[k].detach() - sd0[k]).abs().max()) for k in sd0)
    assert moved > 5e-3                                   # the fixture is sensitive to the updates
    from tests.golden.recipes import check_state
    slack = 0.25 * g["lr"] * len(g["xs"])
    check_state(st.params, g["params"], 5e-4, "param", adam_slack=slack)
    check_state(st.shadow, g["shadow"], 5e-4, "shadow", adam_slack=slack)


def test_g8_toy(golden):
    g = golden("g8_toy.pt")
    assert g["nparams"] == 67074
    sd = _leaf(g["sd"])
    y = toy_ref.decoder_forward(sd, g["x"], g["t"], 128, 3)
    check(y, g["y"], RT, name="toy.y")
    (y * g["gy"]).sum().backward()
    for k, v in g["grads"].items():
        check(sd[k].grad, v, 5e-5, name="toy.grad." + k)
    # 10 training steps of BASELINE config 1 (T=100, beta 1e-3..0.2, B=1000): loss must go down
    gen = torch.Generator().manual_seed(g["data_seed"])
    ang = torch.randint(8, (10, 1000), generator=gen).double() * (3.141592653589793 / 4)
    data = (torch.stack([ang.cos(), ang.sin()], -1) * 2 + 0.1 * torch.randn(10, 1000, 2, generator=gen, dtype=torch.float64)).float()
    ts = torch.randint(100, (10, 1000), generator=gen)
    noises = torch.randn(10, 1000, 2, generator=gen)
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-3, 0.2, 100), "fixed-large")
    p = _leaf(g["sd"])
    opt = torch.optim.Adam(list(p.values()), lr=1e-3)
    losses = []
    for i in range(10):
        x_t = D.q_sample(T, data[i], ts[i], noises[i])
        loss = D.mse_eps_loss(toy_ref.decoder_forward(p, x_t, ts[i], 128, 3), noises[i]).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=2e-4), (losses, g["losses"])
    assert losses[-1] < losses[0]
    import re
    for k, v in g["sd_after"].items():           # temp_fc.{1,2}.* alias temp_fc.0.* (toy_model.py:47-48)
        check(p[re.sub(r"temp_fc\.\d+\.", "temp_fc.0.", k)], v, 2e-3, name="toy.after." + k)
    # sampling plumbing: finite output of the right shape, no clipping (toy/diffusion.py:32)
    with torch.no_grad():
        x_T, zs = _noise_stream(5, (64, 2), 100)
        x = x_T
        for i, ti in enumerate(range(99, -1, -1)):
            t = torch.full((64,), ti, dtype=torch.int64)
            x, _ = D.p_step_from_eps(T, x, t, toy_ref.decoder_forward(p, x, t, 128, 3), zs[i], clip_denoised=False)
        assert x.shape == (64, 2) and bool(torch.isfinite(x).all())
