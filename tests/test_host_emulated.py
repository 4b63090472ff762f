"""Host-logic tests (CPU): the product's engine / diffusion / trainer code runs unmodified, with the C-ABI calls
routed to tests/abi_emulator.py (a numpy restatement of include/ddpm_hip.h over host pointers).  This checks the
orchestration — pitches, zero-copy concat, gradient fan-in, weight caches, autograd wiring, state-dict layout —
against the oracle.  The HIP kernels themselves are checked by the -m gpu tests."""
import json
import os

import pytest
import torch

import ddim as ddim_mod
import ddpm_torch
from ddpm_torch import _hip
from oracle import diffusion_ref as D
from oracle import unet_ref as U
from tests import abi_emulator
from tests.golden.recipes import check, rnd

TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1, apply_attn=[False, True], drop_rate=0.0)
TINY3 = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2, 2], num_res_blocks=2, apply_attn=[False, True, False], drop_rate=0.1)

TINY3_POOL = dict(TINY3, resample_with_conv=False)        # AvgPool2d(2) / bare nearest Upsample between the levels (unet.py:169, :196)

pytestmark = pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")


@pytest.fixture
def emu(monkeypatch):
    return abi_emulator.install(monkeypatch, _hip)


def make(cfg, seed=5, dtype=torch.float32):
    torch.manual_seed(seed)
    m = ddpm_torch.UNet(**cfg)
    sd = U.randomize_state_dict(m.state_dict(), 17)
    m.load_state_dict(sd)
    m.set_compute_dtype(dtype)
    return m, sd


@pytest.mark.parametrize("cfg", [TINY, TINY3, TINY3_POOL], ids=["tiny", "tiny3", "tiny3_pool_resample"])
def test_forward_fp32_matches_oracle(emu, cfg):
    m, sd = make(cfg)
    m.eval()
    x, t = rnd(2, 3, 16, 16, seed=1), torch.tensor([3, 977])
    with torch.no_grad():
        y = m(x, t)
        ref = U.unet_forward(sd, cfg, x, t)
    check(y, ref, 2e-5, name="fwd")
    assert "ddpm_conv2d_nhwc" in emu.log and "ddpm_groupnorm_silu_fwd" in emu.log


def test_forward_bf16_close(emu):
    m, sd = make(TINY, dtype=torch.bfloat16)
    m.eval()
    x, t = rnd(2, 3, 8, 8, seed=2), torch.tensor([10, 500])
    with torch.no_grad():
        y = m(x, t)
        ref = U.unet_forward(sd, TINY, x, t)
    check(y, ref, 5e-2, name="fwd_bf16")      # bf16 storage of every activation: loose, documented tolerance


@pytest.mark.parametrize("cfg", [TINY, TINY3, TINY3_POOL], ids=["tiny", "tiny3", "tiny3_pool_resample"])
def test_backward_fp32_matches_oracle(emu, cfg):
    m, sd = make(cfg)
    m.train()
    m.engine().debug_keep_tape = True
    x, t, gy = rnd(2, 3, 16, 16, seed=3), torch.tensor([7, 912]), rnd(2, 3, 16, 16, seed=4)
    y = m(x, t)
    (y * gy).sum().backward()
    # the oracle consumes the very dropout masks the engine used
    masks = {}
    if cfg["drop_rate"] > 0:
        eng = m.engine()
        rec = {id(r[1]): r for r in eng.last_tape if r[0] == "res"}
        names = {id(mod): name for name, mod in m.named_modules()}
        for rb_id, r in rec.items():
            h1, seed = r[6], r[9]
            n = h1.B * h1.H * h1.W * h1.C
            mk = torch.empty(n)
            emu.ddpm_dropout_mask(mk.data_ptr(), n, cfg["drop_rate"], seed, 0)
            masks[names[rb_id] + "."] = mk.reshape(h1.B, h1.H, h1.W, h1.C).permute(0, 3, 1, 2)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = U.unet_forward(p, cfg, x, t, training=True, masks=masks)
    check(y, ref, 2e-5, name="train fwd")
    (ref * gy).sum().backward()
    for k, prm in m.named_parameters():
        check(prm.grad, p[k].grad, 2e-4, atol=2e-5, name="grad." + k)


def test_backward_slab_split_k_path_gives_the_same_gradients(emu, monkeypatch):
    """DDPM_WGRAD_SLABS=1: split-K slices store into slab copies, ddpm_wgrad_reduce sums them — same gradients as the atomics."""
    import ddpm_torch.models.unet as unet_mod
    x, t, gy = rnd(2, 3, 16, 16, seed=3), torch.tensor([7, 912]), rnd(2, 3, 16, 16, seed=4)
    grads = []
    for slabs in (False, True):
        monkeypatch.setattr(unet_mod, "_WGRAD_SLABS", slabs)
        m, _ = make(TINY)
        m.train()
        eng = m.engine()
        monkeypatch.setattr(eng, "_splits", lambda M, N, K: 3)            # force several slices on the tiny net
        emu.log.clear()
        (m(x, t) * gy).sum().backward()
        assert ("ddpm_wgrad_reduce" in emu.log) == slabs
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    for k in grads[0]:
        check(grads[1][k], grads[0][k], 1e-6, atol=1e-7, name="slab grad." + k)


def test_state_dict_roundtrip_and_cache_refresh(emu):
    m, sd = make(TINY)
    m.eval()
    x, t = rnd(1, 3, 8, 8, seed=6), torch.tensor([42])
    with torch.no_grad():
        y0 = m(x, t)
        sd2 = U.randomize_state_dict(sd, 99)
        m.load_state_dict(sd2)                      # in-place parameter writes must invalidate the packed copies
        y1 = m(x, t)
        ref = U.unet_forward(sd2, TINY, x, t)
    check(y1, ref, 2e-5, name="after load")
    assert float((y1 - y0).abs().max()) > 1e-3


def test_diffusion_kernels_and_loops(emu, golden):
    g = golden("g5_steps.pt")
    betas = ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    for vt in ("fixed-small", "fixed-large"):
        r = g[vt]
        dif = ddpm_torch.GaussianDiffusion(betas, "eps", vt, "mse")
        check(dif.q_sample(r["x0"], r["t"], r["noise"]), r["x_t"], 1e-6, name="x_t")
        lin = lambda x, t: 0.1 * x + 0.01 * t.reshape(-1, 1, 1, 1).to(x)
        check(dif.train_losses(lin, r["x0"], r["t"], noise=r["noise"]), r["loss_lin"], 1e-5, name="loss")
        xp, px0 = dif._step(r["x_t"], lin(r["x_t"], r["t"]), r["z"], r["t"], True, True)
        check(xp, r["x_prev_lin"], 1e-5, name="x_prev")
        check(px0, r["pred_x0_lin"], 1e-5, name="pred_x0")
        mean, var, logvar, _ = dif.p_mean_var(lin, r["x_t"], r["t"], True, True)
        check(mean, r["mean_lin"], 1e-5, name="mean")
        check(logvar, r["logvar"], 1e-6, name="logvar")
    # DDIM: the model is evaluated at subsequence[t]
    sub = ddim_mod.get_selection_schedule("linear", 50, 1000)
    dd = ddim_mod.DDIM.from_ddpm(ddpm_torch.GaussianDiffusion(betas, "eps", "fixed-small", "mse"), eta=0.0, subsequence=sub)
    seen = []
    dd.p_sample(lambda x, t: (seen.append(int(t[0])), 0.1 * x)[1], shape=(2, 3, 4, 4), device="cpu", seed=3)
    assert seen == list(range(980, -1, -20))


def test_sampling_loop_matches_oracle_with_reference_noise_stream(emu, golden):
    g6, g3 = golden("g6_loops.pt"), golden("g3_model.pt")
    torch.manual_seed(g3["tiny_init_seed"])
    m = ddpm_torch.UNet(**g3["tiny_cfg"])
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), g3["tiny_rand_seed"]))
    m.eval()
    betas = ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    sub = ddim_mod.get_selection_schedule("linear", 50, 1000)
    dd = ddim_mod.DDIM(betas, "eps", "fixed-small", "mse", eta=0.0, subsequence=sub)
    r = g6["ddim_linear_50_eta0.0"]
    x = dd.p_sample(m, shape=tuple(r["shape"]), device="cpu", seed=r["seed"])      # CPU generator == the reference's stream
    check(x, r["x_0"], 5e-4, name="ddim50")


def test_trainer_steps_match_reference_fixture(emu, golden):
    g = golden("g7_train.pt")
    torch.manual_seed(g["init_seed"])
    m = ddpm_torch.UNet(**g["cfg"])
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), g["rand_seed"]))
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=g["lr"], betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: min((s + 1) / g["warmup"], 1.0))
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0, shape=(3, 8, 8),
                            device=torch.device("cpu"), ema_decay=0.9999)
    m.train()
    losses = []
    for i, x in enumerate(g["xs"]):
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=2e-5)
    assert tr.ema.num_updates == g["num_updates"]
    for k, v in g["params"].items():
        check(m.state_dict()[k], v, 2e-5, name="param." + k)
    for k, v in g["shadow"].items():
        check(tr.ema.shadow[k], v, 2e-5, name="shadow." + k)


def test_cpu_tensors_fail_loudly_without_emulation():
    m = ddpm_torch.UNet(**TINY)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 8, 8), torch.zeros(1, dtype=torch.int64))
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 10), "eps", "fixed-small", "mse")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dif.q_sample(torch.zeros(1, 3, 4, 4), torch.zeros(1, dtype=torch.int64), torch.zeros(1, 3, 4, 4))


def test_reference_fixtures_through_the_abi_harness(emu, golden):
    """The G1 / G2 / G5 cases of tests/fixture_cases.py with the ABI emulated: checks the harness the GPU run uses."""
    from tests import fixture_cases as FC
    FC.run_g1(golden("g1_ops.pt"), "cpu", dt=0)
    FC.run_g2(golden("g2_blocks.pt"), "cpu")
    FC.run_g5(golden("g5_steps.pt"), golden("g3_model.pt"), "cpu")


def test_product_tables_equal_the_reference_tables(golden):
    """Every attribute the reference's GaussianDiffusion / DDIM keep (fp64), compared to the G4 fixture directly."""
    g = golden("g4_tables.pt")
    names = ("betas", "alphas_bar", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar", "sqrt_recip_alphas_bar", "sqrt_recip_m1_alphas_bar",
             "posterior_var", "posterior_logvar_clipped", "posterior_mean_coef1", "posterior_mean_coef2")
    extra = ("fixed_model_var", "fixed_model_logvar", "alphas", "alphas_bar_prev", "sqrt_alphas_bar_prev")

    def compare(obj, rec, tag):
        for n in names + extra:
            if n in rec and torch.is_tensor(rec[n]):
                got = getattr(obj, n)
                assert got.dtype == torch.float64 and got.shape == rec[n].shape, (tag, n)
                assert float((got - rec[n]).abs().max()) <= 1e-12 * max(1.0, float(rec[n].abs().max())), (tag, n)
        for n in ("model_mean_type", "model_var_type", "loss_type", "timesteps"):
            assert getattr(obj, n) == rec[n], (tag, n)

    lin = ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    for vt in ("fixed-small", "fixed-large"):
        compare(ddpm_torch.GaussianDiffusion(lin, "eps", vt, "mse"), g["ddpm_" + vt], "ddpm_" + vt)
    compare(ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-3, 0.2, 100), "eps", "fixed-large", "mse"), g["toy_fixed-large"], "toy")
    for sched in ("quad", "warmup10", "warmup50", "const", "jsd"):
        assert torch.equal(ddpm_torch.get_beta_schedule(sched, 1e-4, 0.02, 1000), g["betas_" + sched]), sched
    for sched, size in (("linear", 50), ("quadratic", 100)):
        sub = ddim_mod.get_selection_schedule(sched, size, 1000)
        assert torch.equal(sub, g[f"sel_{sched}_{size}"])
        for eta in (0.0, 1.0):
            for vt in ("fixed-small", "fixed-large"):
                compare(ddim_mod.DDIM(lin, "eps", vt, "mse", eta=eta, subsequence=sub), g[f"ddim_{sched}_{size}_eta{eta}_{vt}"], f"ddim {sched} {eta} {vt}")
    assert torch.equal(ddim_mod.get_selection_schedule("quadratic", 50, 1000), g["sel_quadratic_50"])


def test_backward_routes_3x3_weight_gradients_through_the_patch_kernel(emu, monkeypatch):
    """bf16 training: 3x3 / stride-1 weight gradients (and conv2's bias gradient) go through ddpm_conv3x3_wgrad_nhwc with slab
    copies + ddpm_wgrad_reduce; same gradients as the generic path with its separate column sums."""
    import ddpm_torch.models.unet as unet_mod
    x, t, gy = rnd(2, 3, 16, 16, seed=3), torch.tensor([7, 912]), rnd(2, 3, 16, 16, seed=4)
    grads = []
    for patch in (True, False):
        monkeypatch.setattr(unet_mod, "_WGRAD3", patch)
        m, _ = make(TINY, dtype=torch.bfloat16)
        m.train()
        emu.log.clear()
        (m(x, t) * gy).sum().backward()
        assert ("ddpm_conv3x3_wgrad_nhwc" in emu.log) == patch
        assert ("ddpm_wgrad_reduce" in emu.log) == patch
        grads.append(({k: p.grad.clone() for k, p in m.named_parameters()}, emu.log.count("ddpm_colsum")))
    assert grads[0][1] < grads[1][1]                             # folded bias gradients: fewer column-sum launches
    for k in grads[0][0]:
        check(grads[0][0][k], grads[1][0][k], 1e-5, atol=1e-6, name="patch grad." + k)


def test_small_grid_split_plan_is_stable_for_captured_graphs(monkeypatch):
    """Host side of the two-run split of the 4x4 level (_ops.SplitK.plan): offered exactly where the 64x64-tile grid has <= 128 blocks and
    K is long enough; the workspace is reserved once at full size, so the pointers a captured graph holds never move; the switch turns it off."""
    from ddpm_torch import _ops
    sk = _ops.SplitK("cpu")
    s, ws, cnt = sk.plan(128 * 16, 256, 9 * 256, _hip.BF16)                  # CIFAR 4x4 level: 128 tiles of 64x64, 36 K-steps
    assert s == 2 and ws and cnt
    assert sk.plan(128 * 64, 256, 9 * 256, _hip.BF16) == (1, 0, 0)           # 8x8 level: 512 tiles
    assert sk.plan(128 * 16, 256, 512, _hip.BF16) == (1, 0, 0)               # 1x1 shortcut: 8 K-steps
    assert sk.plan(128 * 16, 256, 512, _hip.F32)[0] == 2                     # ... 16 of fp32
    for M, N, K in ((2 * 16, 64, 9 * 512), (100 * 16, 192, 9 * 256), (128 * 16, 256, 9 * 512)):
        assert sk.plan(M, N, K, _hip.BF16) == (2, ws, cnt)                   # same buffers whatever came before
    assert sk.ws.numel() >= 128 * 2 * 16384 and int(sk.cnt.abs().sum()) == 0
    monkeypatch.setattr(_ops, "_SPLITK64", False)
    assert _ops.SplitK("cpu").plan(128 * 16, 256, 9 * 256, _hip.BF16) == (1, 0, 0)


@pytest.mark.parametrize("mean_type", ["eps", "x_0", "mean"])
def test_kl_loss_through_the_product_matches_the_oracle(emu, mean_type):
    """GaussianDiffusion(loss_type="kl").train_losses (diffusion.py:222-224): product wiring (tables, autograd node, fused backward entry) on the
    emulated ABI against the oracle's `loss_term_bpd`, with a batch that mixes t = 0 and t > 0; the gradient reaches the network's parameters."""
    betas = ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    dif = ddpm_torch.GaussianDiffusion(betas, mean_type, "fixed-large", "kl")
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 1000), "fixed-large")
    m, sd = make(TINY)
    m.train()
    x_0 = rnd(3, 3, 8, 8, seed=1).clamp(-1, 1)
    noise, t = rnd(3, 3, 8, 8, seed=2), torch.tensor([0, 7, 912])
    losses = dif.train_losses(m, x_0, t, noise=noise)
    losses.mean().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x_t = D.q_sample(T, x_0, t, noise)
    ref, _ = D.loss_term_bpd(T, mean_type, x_0, x_t, t, U.unet_forward(p, TINY, x_t, t, training=True), clip_denoised=False)
    ref.mean().backward()
    check(losses, ref, 1e-4, name="kl losses")
    scales = {k: float(v.grad.abs().max()) for k, v in p.items()}
    floor = 0.02 * sorted(scales.values())[len(scales) // 2]          # (conv1.bias in front of a GroupNorm: a gradient that is all cancellation)
    worst = max(float((prm.grad - p[k].grad).abs().max()) / max(scales[k], floor) for k, prm in m.named_parameters())
    assert worst < 1e-3, worst
    assert "ddpm_vlb_terms" in emu.log and "ddpm_vlb_terms_bwd" in emu.log
    # evaluation form: clipped estimate returned
    with torch.no_grad():
        l2, pred = dif._loss_term_bpd(m, x_0, x_t, t, clip_denoised=True, return_pred=True)
        r2, rp = D.loss_term_bpd(T, mean_type, x_0, x_t, t, U.unet_forward(sd, TINY, x_t, t, training=True), clip_denoised=True)
    check(l2, r2, 1e-4, name="bpd clipped"); check(pred, rp, 1e-4, name="pred_x0")
