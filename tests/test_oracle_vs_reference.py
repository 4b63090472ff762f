"""CPU, BUILD CONTAINER ONLY (skipped wherever /root/reference is absent — always on the GPU box): the oracle against the live
reference, beyond what the committed fixtures freeze.  Fresh seeds each run of this file would defeat reproducibility, so the seeds
are fixed but DIFFERENT from the ones tests/golden/make_golden.py uses: an oracle that merely memorised the fixtures fails here."""
import pytest
import torch

from oracle import diffusion_ref as D
from oracle import load_reference as LR
from oracle import unet_ref as U

pytestmark = pytest.mark.skipif(not LR.available(), reason="reference checkout not mounted (build container only)")

CFG = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2, 2], num_res_blocks=2, apply_attn=[False, True, False], drop_rate=0.0)


@pytest.fixture(scope="module")
def ref():
    torch.set_num_threads(1)
    return LR.load()


CFG_POOL = dict(CFG, resample_with_conv=False)          # AvgPool2d(2) / bare nearest Upsample (unet.py:163-170,196-199)


def _model(ref, seed, cfg=CFG):
    torch.manual_seed(seed)
    m = ref.UNet(**cfg)
    sd = U.randomize_state_dict(m.state_dict(), seed + 1)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("cfg", [CFG, CFG_POOL], ids=["conv_resample", "pool_resample"])
def test_seeded_init_and_key_order_match(ref, cfg):
    torch.manual_seed(77)
    m = ref.UNet(**cfg)
    torch.manual_seed(77)
    mine = U.init_state_dict(cfg)
    theirs = m.state_dict()
    assert list(mine) == list(theirs)
    for k in mine:
        assert torch.equal(mine[k], theirs[k]), k


@pytest.mark.parametrize("cfg", [CFG, CFG_POOL], ids=["conv_resample", "pool_resample"])
def test_forward_and_every_gradient(ref, cfg):
    m, sd = _model(ref, 901, cfg)
    g = torch.Generator().manual_seed(902)
    x, gy = torch.randn(3, 3, 16, 16, generator=g), torch.randn(3, 3, 16, 16, generator=g)
    t = torch.tensor([0, 412, 999])
    m.train()
    y = m(x, t)
    (y * gy).sum().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y2 = U.unet_forward(p, cfg, x, t, training=True)
    (y2 * gy).sum().backward()
    assert float((y - y2).detach().abs().max()) <= 2e-5 * float(y.detach().abs().max())
    for k, q in m.named_parameters():
        scale = max(float(q.grad.abs().max()), 1e-6)
        assert float((q.grad - p[k].grad).abs().max()) <= 2e-4 * scale + 1e-6, k


@pytest.mark.parametrize("var_type", ["fixed-small", "fixed-large"])
def test_progressive_sampling_loop(ref, var_type):
    """p_sample_progressive (diffusion.py:176-198) on a 60-step chain: final sample and every kept pred_x0 against the oracle's loop
    fed with the noise stream the reference consumed (x_T first, then one z per step)."""
    m, sd = _model(ref, 911)
    m.eval()
    betas = ref.get_beta_schedule("linear", 1e-4, 0.02, 60)
    dif = ref.GaussianDiffusion(betas, "eps", var_type, "mse")
    shape = (2, 3, 16, 16)
    x, preds = dif.p_sample_progressive(m, shape=shape, device=torch.device("cpu"), pred_freq=10, seed=5)
    g = torch.Generator().manual_seed(5)
    x_T = torch.empty(shape).normal_(generator=g)
    zs = [torch.empty(shape).normal_(generator=g) for _ in range(60)]
    T = D.ddpm_tables(D.beta_schedule("linear", 1e-4, 0.02, 60), var_type)
    with torch.no_grad():
        fn = lambda xt, t: U.unet_forward(sd, CFG, xt, t, training=False)     # noqa: E731
        mine = D.sample_loop(T, fn, x_T, zs)
        # pred_x0 of the kept steps, recomputed along the oracle's own trajectory
        xt, kept = x_T, []
        for i, ti in enumerate(range(59, -1, -1)):
            t = torch.full((2,), ti, dtype=torch.int64)
            eps = fn(xt, t)
            nxt, pred = D.p_step_from_eps(T, xt, t, eps, zs[i], clip_denoised=True)
            if (ti + 1) % 10 == 0:
                kept.append(pred)
            xt = nxt
    assert float((mine - x).abs().max()) <= 1e-4 * float(x.abs().max())
    assert preds.shape == (6,) + shape
    for j, pr in enumerate(kept):                    # the reference fills preds from the back: preds[L-1] is the first one kept
        assert float((preds[5 - j] - pr).abs().max()) <= 1e-4, j
