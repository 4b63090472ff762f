"""-m gpu: the BASELINE.json configurations round 1 left unexercised, the bf16 gradient bar on the real CIFAR geometry, and
the captured (hipGraph) training / sampling steps against their eager twins and the reference fixtures."""
import os

import pytest
import torch

import ddim as ddim_mod
import ddpm_torch
from ddpm_torch import _hip
from ddpm_torch.utils import train as train_mod
from oracle import unet_ref as U
from tests.golden.recipes import check, check_state, rnd
from tests.test_unet_gpu import CELEBAHQ, CIFAR, DEV, make, tiny_from_golden

pytestmark = pytest.mark.gpu


def _grad_report(model, ref_params, floor_frac):
    scales = {k: float(v.grad.abs().max()) for k, v in ref_params.items()}
    floor = floor_frac * sorted(scales.values())[len(scales) // 2]
    rows = sorted(((float((q.grad.cpu() - ref_params[k].grad).abs().max()) / max(scales[k], floor), k) for k, q in model.named_parameters()), reverse=True)
    return rows, scales


def _masks_from_tape(m, drop_rate):
    eng, names, masks = m.engine(), {id(mod): name for name, mod in m.named_modules()}, {}
    for rec in eng.last_tape:
        if rec[0] != "res":
            continue
        h1, seed = rec[6], rec[9]
        n = h1.B * h1.H * h1.W * h1.C
        mk = torch.empty(n, device=DEV)
        _hip.call("ddpm_dropout_mask", mk.data_ptr(), n, drop_rate, seed, _hip.stream())
        masks[names[id(rec[1])] + "."] = mk.cpu().reshape(h1.B, h1.H, h1.W, h1.C).permute(0, 3, 1, 2)
    return masks


def test_celebahq_256_fp32_forward_backward_vs_oracle():
    """configs/celebahq.json at its real resolution: K = 9216 reductions, 1024-channel GroupNorms, 65536-pixel images per
    sample and the 512-channel attention — forward <= 1e-3, every parameter gradient <= 3e-3 (fp32 mode)."""
    m, sd = make(CELEBAHQ, dtype=torch.float32)
    m.train()
    x, t, gy = rnd(1, 3, 256, 256, seed=1), torch.tensor([417]), rnd(1, 3, 256, 256, seed=2)
    y = m(x.to(DEV), t.to(DEV))
    (y * gy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = U.unet_forward(p, CELEBAHQ, x, t, training=True)
    (ref * gy).sum().backward()
    rel = float((y.detach().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    rows, scales = _grad_report(m, p, 0.02)
    print(f"celebahq 256 fp32: fwd rel {rel:.3e}; worst grads " + ", ".join(f"{k}={e:.2e}" for e, k in rows[:4]))
    assert rel < 1e-3
    assert rows[0][0] < 3e-3, rows[:4]


def test_celebahq_256_bf16_train_step_properties():
    """BASELINE config 5 per-GPU work (B = 2, bf16, dropout 0): a full Trainer.step twice — finite loss of the right size,
    every parameter finite and moved, EMA shadow follows, and the step is reproducible from the same seeds."""
    finals = []
    for rep in range(2):
        m, _ = make(CELEBAHQ, dtype=torch.bfloat16)
        m.train()
        dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-small", "mse")
        opt = torch.optim.Adam(m.parameters(), lr=2e-5)
        tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 256, 256), device=torch.device(DEV))
        x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(DEV)
        before = {k: v.detach().clone() for k, v in m.named_parameters()}
        for i in range(2):
            tr.stats.reset()
            tr.step(x, global_steps=i + 1)
            assert 0.2 < tr.current_stats["loss"] < 5.0, tr.current_stats
        moved = 0
        for k, v in m.named_parameters():
            assert torch.isfinite(v).all(), k
            moved += int(float((v.detach() - before[k]).abs().max()) > 0)
        assert moved >= 0.95 * len(before)
        assert tr.ema.num_updates == 1
        finals.append(tr.current_stats["loss"])
    assert finals[0] == pytest.approx(finals[1], rel=2e-3)          # atomics in the weight gradients: not bit-identical


def test_cifar_geometry_bf16_gradients_vs_oracle():
    """bf16 mode on the real CIFAR network (B = 4, dropout masks injected into the oracle): per-tensor gradient error
    relative to the tensor's largest gradient <= 5e-2 (floor: 10 % of the median tensor scale for the analytically-zero ones)."""
    m, sd = make(CIFAR, dtype=torch.bfloat16)
    m.train()
    m.engine().debug_keep_tape = True
    x, t, gy = rnd(4, 3, 32, 32, seed=3), torch.tensor([7, 912, 300, 650]), rnd(4, 3, 32, 32, seed=4)
    y = m(x.to(DEV), t.to(DEV))
    (y * gy.to(DEV)).sum().backward()
    masks = _masks_from_tape(m, CIFAR["drop_rate"])
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = U.unet_forward(p, CIFAR, x, t, training=True, masks=masks)
    (ref * gy).sum().backward()
    rel = float((y.detach().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    rows, scales = _grad_report(m, p, 0.1)
    import statistics
    med = statistics.median(e for e, _ in rows)
    print(f"cifar bf16 grads: fwd rel {rel:.3e}; median tensor err {med:.2e}; worst " + ", ".join(f"{k}={e:.2e}" for e, k in rows[:5]))
    assert rel < 4e-2
    assert rows[0][0] < 5e-2, rows[:5]


# ----------------------------------------------------------------------------------------------- training step variants
def _g9_trainer(g, dtype=torch.float32):
    torch.manual_seed(g["init_seed"])
    m = ddpm_torch.UNet(**g["cfg"])
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), g["rand_seed"]))
    m.to(DEV).set_compute_dtype(dtype)
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=g["lr"], betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 1.0 if s < 3 else 0.5)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, grad_norm=1.0, shape=(3, 8, 8),
                            device=torch.device(DEV), ema_decay=0.9999)
    m.train()
    return m, opt, sched, tr


def _reference_stream(g):
    gen = torch.Generator("cpu").manual_seed(g["gen_seed"])       # the reference's CPU (t, noise) stream (utils/train.py:115,138-140)

    def fill(t_buf, noise_buf):
        t_buf.copy_(torch.empty(t_buf.shape, dtype=torch.int64).random_(to=1000, generator=gen))
        noise_buf.copy_(torch.empty(noise_buf.shape).normal_(generator=gen))
    return fill


@pytest.mark.parametrize("path", ["direct", "plan", "autograd"])
def test_steps_that_move_the_weights_vs_reference_fixture(golden, monkeypatch, path):
    """G9 on the GPU (fp32 mode): lr 3e-3 without warm-up — losses after the first step depend on every derived weight copy
    being re-derived from the updated parameters.  "plan": the same six steps with the direct step recorded as a launch plan at step 2
    and replayed by csrc/plan.hip from step 3 on (the learning rate changes at step 4: read from the device block)."""
    g = golden("g9_train_lr.pt")
    monkeypatch.setenv("DDPM_TORCH_AMD_DIRECT_STEP", "0" if path == "autograd" else "1")
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", "plan" if path == "plan" else False)
    m, opt, sched, tr = _g9_trainer(g)
    fill = _reference_stream(g)
    if path != "autograd":
        tr.input_source = fill
    else:
        def get_input(x):
            t, noise = torch.empty(x.shape[0], dtype=torch.int64), torch.empty(x.shape)
            fill(t, noise)
            return {"x_0": x.to(DEV), "t": t.to(DEV), "noise": noise.to(DEV)}
        tr.get_input = get_input
    losses = []
    for i, x in enumerate(g["xs"]):
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    if path == "plan":
        ds = next(iter(tr._direct.values()))
        assert ds.plan is not None and ds.last_kind == "plan" and ds.captures == 1
    print(path, losses)
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=1e-3), (losses, g["losses"])
    slack = 0.25 * g["lr"] * len(g["xs"])
    check_state({k: v.cpu() for k, v in m.state_dict().items()}, g["params"], 2e-3, "param", adam_slack=slack)
    check_state({k: v.cpu() for k, v in tr.ema.shadow.items()}, g["shadow"], 2e-3, "shadow", adam_slack=slack)
    assert tr.ema.num_updates == g["num_updates"]
    m.eval()
    x, t = rnd(2, 3, 8, 8, seed=5), torch.tensor([3, 700])
    with torch.no_grad():
        check(m(x.to(DEV), t.to(DEV)).cpu(), U.unet_forward({k: v.cpu() for k, v in m.state_dict().items()}, g["cfg"], x, t), 1e-3, name="fwd after steps")


@pytest.mark.parametrize("cfg_name,dtype", [("tiny3", torch.float32), ("cifar", torch.bfloat16), ("celebahq", torch.bfloat16)])
def test_captured_training_step_equals_the_eager_step(monkeypatch, cfg_name, dtype):
    """The hipGraph-replayed step and the launch-plan step (the same C-ABI calls re-issued by csrc/plan.hip) consume the generator, the dropout seeds, the LR schedule and the bias corrections exactly
    like the eager direct step: after 6 steps (1 eager + capture / recording + replays) losses / parameters / EMA / Adam state agree
    (tolerance = atomic-order noise of the weight gradients), and the replayed forward sees the updated weights."""
    from tests.test_unet_gpu import TINY3
    cfg = {"tiny3": TINY3, "cifar": CIFAR, "celebahq": CELEBAHQ}[cfg_name]
    # (celebahq: BASELINE config 5's per-GPU work, 256 x 256 at B = 2 — the two-launch GroupNorm path, K-run small-grid convs, C = 512 attention)
    hw, B = {"tiny3": (16, 4), "cifar": (32, 8), "celebahq": (256, 2)}[cfg_name]
    runs = []
    for graph in (True, "plan", False):
        monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", graph)
        torch.manual_seed(11)
        m, _ = make(cfg, dtype=dtype)
        m.train()
        dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: min((s + 1) / 4, 1.0))
        tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=True, shape=(3, hw, hw), device=torch.device(DEV))
        xs = [(torch.rand(B, 3, hw, hw, generator=torch.Generator().manual_seed(40 + i)) * 2 - 1).to(DEV) for i in range(6)]
        losses = []
        for i, x in enumerate(xs):
            tr.stats.reset()
            tr.step(x, global_steps=i + 1)
            losses.append(tr.current_stats["loss"])
        ds = next(iter(tr._direct.values()))
        assert (ds.graph is not None) == (graph is True) and not ds.graph_failed
        assert (ds.plan is not None) == (graph == "plan") and not ds.plan_failed
        if graph is True:
            assert ds.graph.launches == 1                        # single GPU: the whole step is ONE graph launch
        if graph == "plan":                                      # ... or one ddpm_plan_run over the recorded calls (csrc/plan.hip)
            assert len(ds.plan.segments) == 1 and ds.plan.launches > 100 and ds.last_kind == "plan" and ds.plan._c is not None
        first = next(iter(m.parameters()))
        runs.append(dict(losses=losses, params={k: v.detach().cpu().clone() for k, v in m.named_parameters()},
                         shadow={k: v.cpu().clone() for k, v in tr.ema.shadow.items()}, step=int(opt.state[first]["step"]),
                         m1=opt.state[first]["exp_avg"].cpu().clone(), lr=sched.get_last_lr()[0], upd=tr.ema.num_updates))
    b = runs[-1]
    for a in runs[:-1]:
        print(cfg_name, a["losses"], b["losses"])
        tol = 2e-4 if dtype == torch.float32 else 3e-2
        assert a["step"] == b["step"] == 6 and a["lr"] == b["lr"] and a["upd"] == b["upd"] == 5
        assert torch.allclose(torch.tensor(a["losses"]), torch.tensor(b["losses"]), rtol=tol)
        assert cfg_name == "celebahq" or a["losses"][-1] < a["losses"][0]      # it trains (six steps of the 114 M-parameter net need not show it)
        for k in a["params"]:
            scale = float(b["params"][k].abs().max()) or 1.0
            assert float((a["params"][k] - b["params"][k]).abs().max()) <= (tol * scale + 6 * 1e-3 * 0.3), k
        check(a["m1"], b["m1"], tol * 10, atol=1e-6, name="exp_avg")


def test_sampler_graph_is_cached_and_follows_weight_changes(golden, monkeypatch):
    """Second p_sample with the same (model, shape) replays the cached graph (no new capture); swapping the EMA weights in
    (in-place parameter writes) is picked up by the eager refresh before the replay: results equal the eager loop's."""
    m, _ = tiny_from_golden(golden("g3_model.pt"))
    m.eval()
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 60), "eps", "fixed-large", "mse")
    captures = []
    orig = dif._capture_sample_step
    monkeypatch.setattr(dif, "_capture_sample_step", lambda *a, **k: (captures.append(1), orig(*a, **k))[1])
    a1 = dif.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=5)
    a2 = dif.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=5)
    a3 = dif.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=6)
    assert len(captures) == 1 and torch.equal(a1, a2) and not torch.equal(a1, a3)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)                                         # what `with ema:` does: in-place writes through the parameters
    b_graph = dif.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=5)
    monkeypatch.setenv("DDPM_TORCH_AMD_GRAPH", "0")
    b_eager = dif.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=5)
    assert len(captures) == 1
    assert torch.equal(b_graph, b_eager) and not torch.equal(b_graph, a1)
    # injected x_T (the `noise=` argument) through the cached graph
    monkeypatch.setenv("DDPM_TORCH_AMD_GRAPH", "1")
    xT = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(3)).to(DEV)
    c_graph = dif.p_sample(m, noise=xT, device=DEV, seed=9)
    monkeypatch.setenv("DDPM_TORCH_AMD_GRAPH", "0")
    c_eager = dif.p_sample(m, noise=xT, device=DEV, seed=9)
    assert torch.equal(c_graph, c_eager)


def test_sampler_time_table_follows_weights_and_schedule_length(golden, monkeypatch):
    """The samplers take the per-block time biases from a [T][sum Cout] table of all timesteps (one gather per step instead of the
    embedding MLP): same samples as with the MLP run every step (DDPM_TIME_TABLE=0) up to the fp32 summation order of the two GEMM
    shapes; a sampler with a LONGER schedule re-allocates the table, after which the first sampler must not replay a step captured
    against the old one — checked across an in-place weight change in between."""
    m, _ = tiny_from_golden(golden("g3_model.pt"))
    m.eval()
    mk = lambda T: ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, T), "eps", "fixed-large", "mse")
    d20, d40 = mk(20), mk(40)
    shape = (2, 3, 8, 8)

    def both_ways(dif, seed):
        monkeypatch.setenv("DDPM_TIME_TABLE", "1")
        a = dif.p_sample(m, shape=shape, device=DEV, seed=seed)
        monkeypatch.setenv("DDPM_TIME_TABLE", "0")
        b = dif.p_sample(m, shape=shape, device=DEV, seed=seed)
        monkeypatch.setenv("DDPM_TIME_TABLE", "1")
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), float((a - b).abs().max())
        return a

    a20 = both_ways(d20, 3)
    eng = m.engine()
    assert eng.tt_T == 20 and not eng.tt_on                      # the flag is only up inside a sampler
    both_ways(d40, 4)
    assert eng.tt_T == 40
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.02)
    b20 = both_ways(d20, 3)                                      # new weights, table moved: a stale captured step would reproduce a20
    assert not torch.equal(a20, b20)
    # a t outside the table (a schedule longer than any sampler announced) cannot be read silently: rows come back as NaN
    t_bad = torch.full((2,), 45, dtype=torch.int64, device=DEV)
    eng.tt_on = True
    try:
        with torch.no_grad():
            y = m(torch.zeros(shape, device=DEV), t_bad)
    finally:
        eng.tt_on = False
    assert bool(torch.isnan(y).any())


def test_training_steps_after_a_sample_grid_do_not_rebuild_the_time_table(golden):
    """upstream utils/train.py:209-216 samples an image grid between epochs; after it the table's size stays at T, and every
    `Trainer.step` bumps the parameters' versions.  The table belongs to the samplers: the training steps that follow must not run the
    T-row embedding MLP before each replay (nothing in training reads it) — counted here as calls of the engine's MLP with T rows —
    and the next sampler must see the trained weights (one rebuild)."""
    g = golden("g9_train_lr.pt")
    m, opt, sched, tr = _g9_trainer(g)
    tr.input_source = _reference_stream(g)
    dif20 = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 20), "eps", "fixed-large", "mse")
    m.eval()
    a = dif20.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=5)
    eng = m.engine()
    assert eng.tt_T == 20 and not eng.tt_on
    rows = []
    inner = eng._time_biases
    eng._time_biases = lambda t, B: (rows.append(B), inner(t, B))[1]
    try:
        m.train()
        for i, x in enumerate(g["xs"][:4]):                    # eager step, capture, replays
            tr.step(x, global_steps=i + 1)
        torch.cuda.synchronize()
        assert 20 not in rows, rows                            # no T-row pass while training
        m.eval()
        b = dif20.p_sample(m, shape=(2, 3, 8, 8), device=DEV, seed=5)
        assert rows.count(20) == 1, rows                       # the sampler rebuilt it once, for the new weights
    finally:
        eng._time_biases = inner
    assert torch.isfinite(b).all() and not torch.equal(a, b)


def test_ddim50_celeba_quadratic_eta1_vs_eager(monkeypatch):
    """BASELINE config 4 network at 64x64 (B = 2): DDIM (quadratic, eta = 1: noise is consumed) graph == eager, finite."""
    from tests.test_unet_gpu import CELEBA
    m, _ = make(CELEBA, dtype=torch.bfloat16)
    m.eval()
    betas = ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000)
    for sched, eta in (("linear", 0.0), ("quadratic", 1.0)):
        dd = ddim_mod.DDIM(betas, "eps", "fixed-small", "mse", eta=eta, subsequence=ddim_mod.get_selection_schedule(sched, 50, 1000))
        monkeypatch.setenv("DDPM_TORCH_AMD_GRAPH", "1")
        a = dd.p_sample(m, shape=(2, 3, 64, 64), device=DEV, seed=131071)
        monkeypatch.setenv("DDPM_TORCH_AMD_GRAPH", "0")
        b = dd.p_sample(m, shape=(2, 3, 64, 64), device=DEV, seed=131071)
        assert torch.isfinite(a).all() and torch.equal(a, b), (sched, float((a - b).abs().max()))


def test_auto_mode_picks_a_step_execution_and_keeps_training(monkeypatch):
    """Default (auto): a few eager, a few launch-plan and a few graph-replayed steps are timed, one form is kept; the loss keeps decreasing through the
    hand-over and both probes were real training steps."""
    from tests.test_unet_gpu import TINY3
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", "auto")
    torch.manual_seed(3)
    m, _ = make(TINY3, dtype=torch.float32)
    m.train()
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 16, 16), device=torch.device(DEV))
    x = (torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    losses = []
    for i in range(20):
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    ds = next(iter(tr._direct.values()))
    assert ds.choice in ("eager", "plan", "graph") and set(ds.times) == {"eager", "plan", "graph"} and min(ds.times.values()) > 0
    assert not ds.graph_failed and not ds.plan_failed and ds.settled() and ds.last_kind == ds.choice
    assert (ds.graph is not None) == (ds.choice == "graph") and (ds.plan is not None) == (ds.choice == "plan")     # the losers' pools are released
    assert int(opt.state[next(iter(m.parameters()))]["step"]) == 20 and tr.ema.num_updates == 19
    assert sum(losses[-4:]) < sum(losses[:4])


def test_captured_step_is_recaptured_when_baked_addresses_move(monkeypatch):
    """A captured step holds raw addresses (the fused update's pointer table, the engine's packed weights and workspaces).
    `optimizer.load_state_dict` re-creates the Adam moments (new table), `model.float()` / `.to()` drops the engine: the next
    step must notice, capture again and give what the eager step gives — never replay through the stale pointers."""
    from tests.test_unet_gpu import TINY3
    runs = {}
    for graph in (True, "plan", False):
        monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", graph)
        torch.manual_seed(11)
        m, _ = make(TINY3, dtype=torch.float32)
        m.train()
        dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 16, 16), device=torch.device(DEV))
        xs = [(torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(70 + i)) * 2 - 1).to(DEV) for i in range(9)]
        for i in range(3):
            tr.step(xs[i], global_steps=i + 1)
        ds = next(iter(tr._direct.values()))
        assert ds.captures == (1 if graph else 0)
        # (1) the moments are re-created: every exp_avg / exp_avg_sq tensor is a new allocation
        import copy
        state = copy.deepcopy(opt.state_dict())                  # what torch.load of a checkpoint hands over: tensors of its own
        old_ptr = opt.state[next(iter(m.parameters()))]["exp_avg"].data_ptr()
        opt.load_state_dict(state)
        assert opt.state[next(iter(m.parameters()))]["exp_avg"].data_ptr() != old_ptr
        for i in range(3, 6):
            tr.step(xs[i], global_steps=i + 1)
        assert ds.captures == (2 if graph else 0)
        # (2) the engine is re-created (parameter storage unchanged, but every derived buffer is new)
        serial = m.engine().serial
        m._apply(lambda t: t)
        assert m.engine().serial != serial
        for i in range(6, 9):
            tr.step(xs[i], global_steps=i + 1)                   # (the new engine's first step runs eagerly, the next one captures)
        assert ds.captures == (3 if graph else 0) and not ds.graph_failed
        torch.cuda.synchronize()
        runs[graph] = ({k: v.detach().cpu().clone() for k, v in m.named_parameters()}, tr.current_stats["loss"])
    for form in (True, "plan"):
        for k, v in runs[form][0].items():
            scale = float(runs[False][0][k].abs().max()) or 1.0
            assert float((v - runs[False][0][k]).abs().max()) <= 2e-4 * scale + 9 * 1e-3 * 0.3, (form, k)
        assert abs(runs[form][1] - runs[False][1]) <= 2e-4 * abs(runs[False][1])


@pytest.mark.parametrize("form", [True, "plan"])
def test_training_graph_survives_a_sampling_pass_at_another_batch_size(monkeypatch, form):
    """Trainer.train() samples a grid between epochs (another batch size -> another GroupNorm workspace requirement); the captured
    training step still points at the workspace it was captured with, which therefore must stay alive and in place."""
    from tests.test_unet_gpu import TINY3
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", form)
    torch.manual_seed(5)
    m, _ = make(TINY3, dtype=torch.float32)
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 20), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 16, 16), device=torch.device(DEV))
    x = (torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    m.train()
    for i in range(3):
        tr.step(x, global_steps=i + 1)
    ws_before = m.engine()._ws.data_ptr()
    m.eval()
    s = tr.sample_fn(sample_size=16, sample_seed=3)                  # B = 16 > 2: the workspace has to grow
    assert s.shape == (16, 3, 16, 16) and torch.isfinite(s).all()
    eng = m.engine()
    assert eng._ws.data_ptr() != ws_before and any(w is not None and w.data_ptr() == ws_before for w in eng._ws_retired)
    m.train()
    for i in range(3, 6):
        tr.step(x, global_steps=i + 1)
    torch.cuda.synchronize()
    assert next(iter(tr._direct.values())).captures == 1 and all(torch.isfinite(p).all() for p in m.parameters())


@pytest.mark.parametrize("cfg_name,hw,B", [("cifar", 32, 8), ("celebahq", 128, 2)])
def test_recorded_step_contains_no_torch_arithmetic_on_the_device(monkeypatch, cfg_name, hw, B):
    """The launch plan replays what went through the C ABI and nothing else (ddpm_torch/_plan.py): on the device — side stream, bf16 kernels,
    flash attention, the fused update — the recorded body may issue no ATen operation beyond allocations and views.  (The CPU twin of
    this test covers the emulated path: tests/test_trainer_host.py.)"""
    import collections
    from torch.utils._python_dispatch import TorchDispatchMode
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", "plan")
    torch.manual_seed(2)
    m, _ = make({"cifar": CIFAR, "celebahq": CELEBAHQ}[cfg_name], dtype=torch.bfloat16)
    m.train()
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, grad_norm=1.0, shape=(3, hw, hw), device=torch.device(DEV))
    x = (torch.rand(B, 3, hw, hw, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(DEV)
    tr.step(x, global_steps=1)
    ds = next(iter(tr._direct.values()))
    seen = collections.Counter()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen[str(func)] += 1
            return func(*args, **(kwargs or {}))

    body = ds.body

    def logged(cut=None, draw=True):
        with Log():
            return body(cut, draw)
    ds.body = logged
    tr.step(x, global_steps=2)                                   # the recording step
    ds.body = body
    tr.step(x, global_steps=3)                                   # a replay
    torch.cuda.synchronize()
    assert ds.plan is not None and ds.last_kind == "plan" and ds.plan.launches > 300
    allowed = {"aten.empty.memory_format", "aten.empty_like.default", "aten.empty_strided.default", "aten.select.int", "aten.view.default",
               "aten.slice.Tensor", "aten.as_strided.default", "aten.detach.default", "aten.alias.default", "aten._unsafe_view.default"}
    assert set(seen) <= allowed, sorted(set(seen) - allowed)
    assert all(torch.isfinite(p).all() for p in m.parameters())


def test_recording_a_plan_while_an_earlier_one_awaits_the_garbage_collector(monkeypatch):
    """A trainer and its step object reference each other, so a dropped trainer's launch plan (and the allocator pool it owns) is freed by
    the CYCLIC collector, whenever that runs.  Releasing a pool while allocations are routed to another one — i.e. in the middle of the next
    recording or capture — aborts the process inside the caching allocator (seen in round 5: `Fatal Python error: Aborted ... Garbage-
    collecting` under `LaunchPlan.record`).  `record` / `capture` therefore collect first and keep the collector off while they run: here a
    collection is forced in the MIDDLE of the second recording, with the first trainer left as garbage just before."""
    import gc
    from tests.test_unet_gpu import TINY3
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", "plan")

    def trainer(seed):
        torch.manual_seed(seed)
        m, _ = make(TINY3, dtype=torch.float32)
        m.train()
        dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
        return m, ddpm_torch.Trainer(m, torch.optim.Adam(m.parameters(), lr=1e-3), dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 16, 16),
                                     device=torch.device(DEV))
    x = (torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    gc.collect()
    gc.disable()                                                 # nothing may free the first trainer before the second recording starts
    try:
        m1, t1 = trainer(1)
        for i in range(3):
            t1.step(x, global_steps=i + 1)
        assert next(iter(t1._direct.values())).plan is not None
        torch.cuda.synchronize()
        del m1, t1                                               # garbage now: a cycle that owns a launch plan and its pool
        m2, t2 = trainer(2)
        t2.step(x, global_steps=1)
        ds = None
        orig_fwd = m2.engine().forward

        def forward_with_collection(*a, **k):                    # runs inside the recorded body
            gc.collect()
            return orig_fwd(*a, **k)
        m2.engine().forward = forward_with_collection
        t2.step(x, global_steps=2)                               # records
        m2.engine().forward = orig_fwd
        t2.step(x, global_steps=3)
        torch.cuda.synchronize()
        ds = next(iter(t2._direct.values()))
        assert ds.plan is not None and ds.last_kind == "plan"
    finally:
        gc.enable()


_FAILED_CAPTURE = r"""
import os, sys, warnings
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import ddpm_torch
import ddpm_torch.utils.train as train_mod
from tests.test_unet_gpu import TINY3, make
train_mod._TRAIN_GRAPH = True
torch.manual_seed(3)
m, _ = make(TINY3, dtype=torch.float32)
m.train()
dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
tr = ddpm_torch.Trainer(m, torch.optim.Adam(m.parameters(), lr=1e-3), dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 16, 16), device=torch.device("cuda"))
x = (torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
tr.step(x, global_steps=1)                                   # eager (first step of an engine)
eng = m.engine()
orig = eng._close_backward
def poisoned(*a, **k):
    # the end of the backward: the weight-gradient stream has been forked into the capture and is not joined yet, so the capture cannot
    # even be ended properly (hipStreamEndCapture: unjoined work) — the worst case for the clean-up
    if torch.cuda.is_current_stream_capturing():
        torch.tensor([[1, 2], [3, 4]], dtype=torch.int64, device="cuda")      # a host -> device copy: refused under capture
    return orig(*a, **k)
eng._close_backward = poisoned
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    tr.step(x, global_steps=2)                               # capture fails half-way (t and noise already drawn inside it) -> eager
ds = next(iter(tr._direct.values()))
assert ds.graph_failed and ds.graph is None and ds.last_kind == "eager", (ds.graph_failed, ds.last_kind)
assert any("capture of the training step failed" in str(x_.message) for x_ in w), [str(x_.message)[:80] for x_ in w]
eng._close_backward = orig
for i in range(3, 6):
    tr.step(x, global_steps=i)                               # ... and training goes on
torch.cuda.synchronize()
assert ds.last_kind == "eager" and all(torch.isfinite(p).all() for p in m.parameters())
loss = tr.current_stats["loss"]
assert loss == loss and 0 < loss < 10, loss
print("FALLBACK_OK", flush=True)
"""


def test_a_capture_that_dies_half_way_falls_back_to_eager_steps_and_a_clean_exit():
    """The hipGraph form is optional: when a capture fails the step must run eagerly — in THIS call and afterwards — and the process must
    end normally.  Two things used to stand in the way (seen in round 5 when a capture hit a host -> device copy): the training generator
    stayed in its in-graph mode ("Offset increment outside graph capture encountered unexpectedly" on the next eager draw) and the caching
    allocator still believed a capture was under way (abort on `captures_underway.empty()` at exit).  Run in a subprocess: the exit code
    is part of the contract."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _FAILED_CAPTURE], capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0 and "FALLBACK_OK" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
