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


@pytest.mark.parametrize("mean_type", ["eps", "x_0", "mean"])
@pytest.mark.parametrize("var_type", ["fixed-small", "fixed-large"])
def test_variational_bound_terms_and_their_gradient(ref, mean_type, var_type):
    """loss_type="kl" (diffusion.py:222-224 -> `_loss_term_bpd`, :203-215): the oracle's restatement against the
    live reference on a batch that mixes t = 0 (decoder NLL) with t > 0 (KL), values and the gradient with respect to the network output."""
    betas = D.beta_schedule("linear", 1e-4, 0.02, 1000)
    dif = ref.GaussianDiffusion(betas=betas, model_mean_type=mean_type, model_var_type=var_type, loss_type="kl")
    T = D.ddpm_tables(betas, var_type)
    g = torch.Generator().manual_seed(4242)
    x_0 = (torch.rand(5, 3, 8, 8, generator=g) * 2 - 1).clamp(-1, 1)
    x_0[0, 0, 0, :4] = torch.tensor([1.0, -1.0, 0.9995, -0.9995])          # both open tails of the discretised likelihood
    noise = torch.randn(5, 3, 8, 8, generator=g)
    t = torch.tensor([0, 0, 1, 500, 999])
    w = torch.randn(5, 3, 8, 8, generator=g) * 0.3
    outs = {}

    def net(x_t, tt):                 # a "network" whose output is a leaf we can differentiate with respect to
        o = (0.7 * x_t + w).detach().requires_grad_(True)
        outs["o"] = o
        return o
    losses = dif.train_losses(net, x_0, t, noise=noise)
    losses.sum().backward()
    o_ref = outs["o"]
    x_t = D.q_sample(T, x_0, t, noise)
    o = o_ref.detach().clone().requires_grad_(True)
    mine, _ = D.loss_term_bpd(T, mean_type, x_0, x_t, t, o, clip_denoised=False)
    mine.sum().backward()
    assert torch.allclose(mine.detach(), losses.detach(), rtol=2e-5, atol=1e-6), (mine, losses)
    assert float((o.grad - o_ref.grad).abs().max()) <= 1e-4 * float(o_ref.grad.abs().max()) + 1e-9      # (fp32 cancellation in cdf_upper - cdf_lower)
    # (`_prior_bpd` / `calc_all_bpd`, diffusion.py:245-267, cannot run in the reference: the jit-scripted `normal_kl` rejects the float
    #  arguments `_prior_bpd` passes — nothing to pin, and the product does not carry them)
    with pytest.raises(RuntimeError):
        dif._prior_bpd(x_0)
    # clip_denoised = True (the evaluation path of calc_all_bpd) and the x_0 estimate
    with torch.no_grad():
        l2, p2 = dif._loss_term_bpd(lambda a, b: o_ref.detach(), x_0, x_t, t, clip_denoised=True, return_pred=True)
        m2, q2 = D.loss_term_bpd(T, mean_type, x_0, x_t, t, o_ref.detach(), clip_denoised=True)
    assert torch.allclose(m2, l2, rtol=2e-5, atol=1e-6) and torch.allclose(q2, p2, rtol=1e-5, atol=1e-6)
