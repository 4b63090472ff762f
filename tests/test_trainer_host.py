"""Host-logic tests of Trainer / EMA / checkpoints (CPU; the C-ABI calls are routed to tests/abi_emulator.py).

The G9 fixture (reference Trainer, lr 3e-3, no warm-up) moves the weights by ~1e-2 per step: reproducing its losses
requires every forward to see the parameters the previous update wrote — through the packed weight copies and the
concatenated time-bias projection the engine derives from them."""
import os

import pytest
import torch

import ddpm_torch
from ddpm_torch import _hip
from ddpm_torch.utils import train as train_mod
from oracle import unet_ref as U
from tests import abi_emulator
from tests.golden.recipes import check, rnd, check_state as _check_state

pytestmark = pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")
TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1, apply_attn=[False, True], drop_rate=0.0)


@pytest.fixture
def emu(monkeypatch):
    return abi_emulator.install(monkeypatch, _hip)


def _g9_trainer(g, use_ema=True):
    torch.manual_seed(g["init_seed"])
    m = ddpm_torch.UNet(**g["cfg"])
    m.load_state_dict(U.randomize_state_dict(m.state_dict(), g["rand_seed"]))
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=g["lr"], betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 1.0 if s < 3 else 0.5)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, scheduler=sched, use_ema=use_ema, grad_norm=1.0, shape=(3, 8, 8),
                            device=torch.device("cpu"), ema_decay=0.9999)
    m.train()
    return m, opt, sched, tr


def _run_g9(tr, g):
    losses = []
    for i, x in enumerate(g["xs"]):
        tr.stats.reset()
        tr.step(x, global_steps=i + 1)
        losses.append(tr.current_stats["loss"])
    return losses


def test_launch_plan_replays_reproduce_the_reference_steps(emu, golden, monkeypatch):
    """The direct step recorded as a launch plan (ddpm_torch/_plan.py; here the table is replayed through the emulator, on the GPU by
    csrc/plan.hip): step 1 eager, step 2 records, steps 3-6 are replays — against the reference Trainer's losses and final state (G9:
    the learning rate changes at step 4, so the replays must read it from the device block, and every forward must see the weights the
    previous replay wrote)."""
    g = golden("g9_train_lr.pt")
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", "plan")
    m, opt, sched, tr = _g9_trainer(g)
    losses = _run_g9(tr, g)
    ds = next(iter(tr._direct.values()))
    assert ds.plan is not None and ds.last_kind == "plan" and ds.captures == 1 and ds.plan.launches > 100
    names = {n for seg in ds.plan.segments for n, _ in seg}
    assert {"ddpm_fill_zero", "ddpm_q_sample", "ddpm_mse_bwd", "ddpm_mt_adam_ema", "ddpm_pack_weight_multi"} <= names
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=5e-4), (losses, g["losses"])
    slack = 0.25 * g["lr"] * len(g["xs"])
    _check_state(m.state_dict(), g["params"], 1e-3, "param", adam_slack=slack)
    _check_state(tr.ema.shadow, g["shadow"], 1e-3, "shadow", adam_slack=slack)
    # ... and bit for bit against the same steps issued eagerly
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", False)
    m2, _, _, tr2 = _g9_trainer(g)
    losses2 = _run_g9(tr2, g)
    assert losses == losses2
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k


def test_recorded_step_contains_no_torch_arithmetic(emu, monkeypatch):
    """A launch plan replays only what went through ``_hip.call``: a torch operation inside the recorded body would run once (at
    recording) and silently be missing from every replay.  Guard: the body of the direct step is run under a TorchDispatchMode while it
    is being recorded — the only ATen calls it may make are allocations and views (the (t, noise) draw sits in front of the recorded
    region; fills and stream edges are C-ABI calls)."""
    import collections
    from torch.utils._python_dispatch import TorchDispatchMode
    monkeypatch.setattr(train_mod, "_TRAIN_GRAPH", "plan")
    torch.manual_seed(0)
    m = ddpm_torch.UNet(**dict(TINY, drop_rate=0.1))
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, grad_norm=1.0, shape=(3, 8, 8), device=torch.device("cpu"))
    m.train()
    x = torch.rand(2, 3, 8, 8) * 2 - 1
    tr.step(x, global_steps=1)                                   # eager warm-up: lazy tables, slabs, workspaces
    ds = next(iter(tr._direct.values()))
    seen = collections.Counter()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen[str(func)] += 1
            return func(*args, **(kwargs or {}))

    monkeypatch.setattr(_hip, "_invoke", lambda name, args: None)     # (the emulator computes with torch: keep it out of the log)
    body = ds.body

    def logged(cut=None, draw=True):
        with Log():
            return body(cut, draw)
    ds.body = logged
    tr.step(x, global_steps=2)                                   # the recording step
    assert ds.plan is not None and ds.plan.launches > 100
    allowed = {"aten.empty.memory_format", "aten.empty_like.default", "aten.empty_strided.default", "aten.select.int", "aten.view.default",
               "aten.slice.Tensor", "aten.as_strided.default", "aten.detach.default", "aten.alias.default", "aten._unsafe_view.default"}
    assert set(seen) <= allowed, sorted(set(seen) - allowed)


@pytest.mark.parametrize("direct", [True, False])
def test_steps_that_move_the_weights_match_the_reference(emu, golden, monkeypatch, direct):
    g = golden("g9_train_lr.pt")
    monkeypatch.setenv("DDPM_TORCH_AMD_DIRECT_STEP", "1" if direct else "0")
    m, opt, sched, tr = _g9_trainer(g)
    losses = _run_g9(tr, g)
    assert ("ddpm_mt_adam_ema" in emu.log)                       # the fused update ran on both paths
    assert ("ddpm_mse_bwd" in emu.log) and (("ddpm_mt_gather_f32" in emu.log))
    assert torch.allclose(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=5e-4), (losses, g["losses"])
    assert tr.ema.num_updates == g["num_updates"]
    assert abs(sched.get_last_lr()[0] - g["last_lr"]) < 1e-15
    assert int(opt.state[next(iter(m.parameters()))]["step"]) == len(g["xs"])
    slack = 0.25 * g["lr"] * len(g["xs"])
    _check_state(m.state_dict(), g["params"], 1e-3, "param", adam_slack=slack)
    _check_state(tr.ema.shadow, g["shadow"], 1e-3, "shadow", adam_slack=slack)
    # the forward after the last update equals the oracle on the NEW state dict (stale derived copies would not)
    m.eval()
    x, t = rnd(2, 3, 8, 8, seed=5), torch.tensor([3, 700])
    with torch.no_grad():
        check(m(x, t), U.unet_forward(m.state_dict(), g["cfg"], x, t), 2e-5, name="fwd after steps")


def test_raw_pointer_update_invalidates_version_keyed_caches(emu, golden, monkeypatch):
    """Regression for the round-1 bug: the fused kernel writes parameters through raw pointers; without the explicit
    version bump the packed conv weights were never re-derived."""
    g = golden("g9_train_lr.pt")
    monkeypatch.setenv("DDPM_TORCH_AMD_DIRECT_STEP", "0")
    m, opt, sched, tr = _g9_trainer(g)
    p0 = m.in_conv.weight
    v0 = p0._version
    tr.step(g["xs"][0], global_steps=1)
    assert p0._version > v0
    emu.log.clear()
    m.eval()
    with torch.no_grad():
        m(rnd(1, 3, 8, 8, seed=1), torch.tensor([1]))
    assert "ddpm_pack_weight_multi" in emu.log and "ddpm_mt_gather_f32" in emu.log


def test_fused_update_follows_reloaded_optimizer_and_ema_state(emu, golden, monkeypatch):
    """optimizer.load_state_dict re-creates the moment tensors: the pointer table must follow them."""
    g = golden("g9_train_lr.pt")
    m, opt, sched, tr = _g9_trainer(g)
    for i in range(2):
        tr.step(g["xs"][i], global_steps=i + 1)
    osd = {"state": {k: {n: (t.clone() if torch.is_tensor(t) else t) for n, t in st.items()} for k, st in opt.state_dict()["state"].items()},
           "param_groups": opt.state_dict()["param_groups"]}
    esd = {"decay": tr.ema.decay, "num_updates": tr.ema.num_updates, "shadow": {k: v.clone() for k, v in reversed(list(tr.ema.shadow.items()))}}
    opt.load_state_dict(osd)
    tr.ema.load_state_dict(esd)                          # keys in another order: matched by name, copied in place
    for i in range(2, len(g["xs"])):
        tr.stats.reset()
        tr.step(g["xs"][i], global_steps=i + 1)
    first = next(iter(m.parameters()))
    assert int(opt.state[first]["step"]) == len(g["xs"])
    slack = 0.25 * g["lr"] * len(g["xs"])
    _check_state(m.state_dict(), g["params"], 1e-3, "param", adam_slack=slack)
    _check_state(tr.ema.shadow, g["shadow"], 1e-3, "shadow", adam_slack=slack)
    assert tr.ema.num_updates == g["num_updates"]


def test_checkpoint_round_trip_and_ddp_prefixes(emu, golden, tmp_path):
    g = golden("g9_train_lr.pt")
    m, opt, sched, tr = _g9_trainer(g)
    for i in range(3):
        tr.step(g["xs"][i], global_steps=i + 1)
    path = str(tmp_path / "ddpm_tiny.pt")
    tr.save_checkpoint(path, epoch=7, loss=0.5)
    saved = str(tmp_path / "ddpm_tiny_7.pt")
    assert os.path.exists(saved)
    chk = torch.load(saved, map_location="cpu", weights_only=False)
    assert set(chk) == {"model", "optimizer", "ema", "scheduler", "epoch", "loss"} and chk["epoch"] == 7
    assert list(chk["model"]) == list(m.state_dict()) and set(chk["ema"]) == {"decay", "shadow", "num_updates"}
    # (a) plain reload into a fresh trainer continues identically
    m2, opt2, sched2, tr2 = _g9_trainer(g)
    tr2.load_checkpoint(saved, map_location="cpu")
    assert tr2.start_epoch == 7 and tr2.ema.num_updates == tr.ema.num_updates
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v)
    for k, v in tr.ema.shadow.items():
        assert torch.equal(tr2.ema.shadow[k], v)
    tr.generator.manual_seed(99); tr2.generator.manual_seed(99)
    tr.stats.reset(); tr2.stats.reset()
    tr.step(g["xs"][3], global_steps=4); tr2.step(g["xs"][3], global_steps=4)
    assert tr.current_stats["loss"] == pytest.approx(tr2.current_stats["loss"], rel=1e-6)
    _check_state(m2.state_dict(), m.state_dict(), 1e-6, "continued")
    # (b) a checkpoint written by a DDP-wrapped run carries 'module.' on the model and shadow keys (utils/train.py:256-262)
    chk["model"] = {"module." + k: v for k, v in chk["model"].items()}
    chk["ema"]["shadow"] = {"module." + k: v for k, v in chk["ema"]["shadow"].items()}
    ddp_path = str(tmp_path / "ddp_style.pt")
    torch.save(chk, ddp_path)
    m3, opt3, sched3, tr3 = _g9_trainer(g)
    tr3.load_checkpoint(ddp_path, map_location="cpu")
    for k, v in chk["model"].items():
        assert torch.equal(m3.state_dict()[k[len("module."):]], v)
    # (c) what generate.py does (generate.py:72-93): prefer the EMA shadow, strip prefixes, freeze
    m4 = ddpm_torch.UNet(**g["cfg"])
    sd = chk["ema"]["shadow"]
    m4.load_state_dict({k[len("module."):]: v for k, v in sd.items()})
    for p in m4.parameters():
        p.requires_grad_(False)
    m4.eval()
    x, t = rnd(2, 3, 8, 8, seed=9), torch.tensor([11, 800])
    ref_sd = {k[len("module."):]: v for k, v in sd.items()}
    check(m4(x, t), U.unet_forward(ref_sd, g["cfg"], x, t), 2e-5, name="ema sample model")


def test_ema_context_swaps_weights_in_and_out(emu, golden):
    g = golden("g9_train_lr.pt")
    m, opt, sched, tr = _g9_trainer(g)
    tr.step(g["xs"][0], global_steps=1)
    live = {k: v.clone() for k, v in m.state_dict().items()}
    x, t = rnd(1, 3, 8, 8, seed=2), torch.tensor([500])
    m.eval()
    with torch.no_grad():
        with tr.ema:
            for k, v in tr.ema.shadow.items():
                assert torch.equal(m.state_dict()[k], v)
            y_ema = m(x, t)
            check(y_ema, U.unet_forward(tr.ema.shadow, g["cfg"], x, t), 2e-5, name="ema weights")
        for k, v in live.items():
            assert torch.equal(m.state_dict()[k], v)
        check(m(x, t), U.unet_forward(live, g["cfg"], x, t), 2e-5, name="restored weights")


def test_train_loop_dry_run_and_model_wrapper(emu, golden, tmp_path):
    g = golden("g9_train_lr.pt")
    m, opt, sched, tr = _g9_trainer(g)
    tr.trainloader = [(x, torch.zeros(len(x))) for x in g["xs"]]      # (images, labels) batches: labels are dropped
    tr.dry_run, tr.num_samples, tr.chkpt_intv = True, 0, 1
    tr.train(chkpt_path=str(tmp_path / "c.pt"))
    assert os.path.exists(str(tmp_path / "c_1.pt"))                   # one step, one epoch, checkpoint written
    assert int(opt.state[next(iter(m.parameters()))]["step"]) == 1
    # ModelWrapper: transforms around the denoiser, '_model.' prefix in its state dict
    w = ddpm_torch.ModelWrapper(m, pre_transform=lambda x: 2 * x, post_transform=lambda y: y + 1)
    assert all(k.startswith("_model.") for k in w.state_dict())
    m.eval()
    x, t = rnd(1, 3, 8, 8, seed=4), torch.tensor([20])
    with torch.no_grad():
        check(w(x, t), m(2 * x, t) + 1, 1e-6, name="wrapper")


def test_input_gradient_flows_through_the_autograd_node(emu):
    """The reference returns d/dx through autograd (models/unet.py:205-233); here in_conv's data gradient — skipped in training — runs when
    x requires grad.  Engine wiring on the emulated ABI against the oracle's autograd (the -m gpu twin: tests/test_unet_gpu.py)."""
    torch.manual_seed(3)
    m = ddpm_torch.UNet(**TINY)
    sd = U.randomize_state_dict(m.state_dict(), 17)
    m.load_state_dict(sd)
    m.train()
    x, t, gy = rnd(2, 3, 8, 8, seed=1), torch.tensor([3, 700]), rnd(2, 3, 8, 8, seed=2)
    xd = x.clone().requires_grad_(True)
    (m(xd, t) * gy).sum().backward()
    xr = x.clone().requires_grad_(True)
    (U.unet_forward({k: v.clone() for k, v in sd.items()}, TINY, xr, t, training=True) * gy).sum().backward()
    check(xd.grad, xr.grad, 2e-4, atol=2e-6, name="d/dx")


def test_running_statistics_and_dummy_scheduler():
    rs = train_mod.RunningStatistics(loss=None)
    rs.update(4, loss=2.0); rs.update(4, loss=6.0)
    assert rs.extract() == {"loss": 1.0}
    rs.reset()
    assert rs.count == 0 and rs.stats == {"loss": 0}
    d = train_mod.DummyScheduler()
    assert d.step() is None and d.state_dict() is None and d.load_state_dict({}) is None
