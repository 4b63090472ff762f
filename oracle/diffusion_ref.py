"""Oracle (test infrastructure): CPU restatement of GaussianDiffusion / DDIM tables and step algebra.

Tables are plain fp64 tensors in a dict; the step functions are pure functions of
(tables, x_t, t, eps_hat, z).  Citations are ``file:line`` in the upstream repo.
Not product code: see ``oracle/__init__.py``.
"""
import math

import torch


def beta_schedule(kind, beta_start, beta_end, timesteps):
    """ddpm_torch/diffusion.py:13-29 (fp64)."""
    f64 = torch.float64
    if kind == "linear":
        b = torch.linspace(beta_start, beta_end, timesteps, dtype=f64)
    elif kind == "quad":
        b = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, timesteps, dtype=f64) ** 2
    elif kind in ("warmup10", "warmup50"):
        frac = 0.1 if kind == "warmup10" else 0.5
        b = beta_end * torch.ones(timesteps, dtype=f64)
        w = int(timesteps * frac)
        b[:w] = torch.linspace(beta_start, beta_end, w, dtype=f64)
    elif kind == "const":
        b = beta_end * torch.ones(timesteps, dtype=f64)
    elif kind == "jsd":
        b = 1.0 / torch.linspace(timesteps, 1, timesteps, dtype=f64)
    else:
        raise NotImplementedError(kind)
    return b


def ddpm_tables(betas, model_var_type):
    """ddpm_torch/diffusion.py:42-73 — every fp64 table the sampler / loss reads."""
    assert betas.dtype == torch.float64
    one = torch.ones(1, dtype=torch.float64)
    alphas = 1.0 - betas
    ab = torch.cumprod(alphas, 0)
    ab_prev = torch.cat([one, ab[:-1]])
    post_var = betas * (1.0 - ab_prev) / (1.0 - ab)
    post_logvar = torch.log(torch.cat([post_var[1:2], post_var[1:]]))      # :65, index 0 := index 1
    T = dict(
        betas=betas, alphas_bar=ab,
        sqrt_alphas_bar=ab.sqrt(), sqrt_one_minus_alphas_bar=(1.0 - ab).sqrt(),
        sqrt_recip_alphas_bar=(1.0 / ab).sqrt(), sqrt_recip_m1_alphas_bar=(1.0 / ab - 1.0).sqrt(),
        posterior_var=post_var, posterior_logvar_clipped=post_logvar,
        posterior_mean_coef1=betas * ab_prev.sqrt() / (1.0 - ab),
        posterior_mean_coef2=alphas.sqrt() * (1.0 - ab_prev) / (1.0 - ab),
    )
    if model_var_type == "fixed-large":                                      # :70-73
        T["fixed_model_var"] = betas
        T["fixed_model_logvar"] = torch.log(torch.cat([post_var[1:2], betas[1:]]))
    elif model_var_type == "fixed-small":
        T["fixed_model_var"] = post_var
        T["fixed_model_logvar"] = post_logvar
    else:
        raise KeyError(model_var_type)
    return T


def selection_schedule(kind, size, timesteps):
    """ddim.py:30-44."""
    if kind == "linear":
        return torch.arange(0, timesteps, timesteps // size)
    if kind == "quadratic":
        return torch.pow(torch.linspace(0, math.sqrt(timesteps * 0.8), size), 2).round().to(torch.int64)
    raise AssertionError(kind)


def ddim_tables(betas, model_var_type, eta, subsequence):
    """ddim.py:47-94 — tables re-derived on alphas_bar[subsequence]."""
    eta2 = eta ** 2
    if eta2 != 1.0 and model_var_type != "fixed-small":
        model_var_type = "fixed-small"                                      # :53-59 silent coercion
    one = torch.ones(1, dtype=torch.float64)
    ab = torch.cumprod(1.0 - betas, 0)[subsequence]
    ab_prev = torch.cat([one, ab[:-1]])
    alphas = ab / ab_prev
    b = 1.0 - alphas
    post_var = b * (1.0 - ab_prev) / (1.0 - ab) * eta2
    post_logvar = torch.log(torch.cat([post_var[1:2], post_var[1:]]).clip(min=1e-20))
    coef2 = torch.sqrt(1 - ab - eta2 * b) * torch.sqrt(1 - ab_prev) / (1.0 - ab)
    coef1 = ab_prev.sqrt() * (1.0 - alphas.sqrt() * coef2)
    T = dict(
        betas=b, alphas=alphas, alphas_bar=ab, alphas_bar_prev=ab_prev,
        sqrt_alphas_bar_prev=ab_prev.sqrt(),
        sqrt_alphas_bar=ab.sqrt(), sqrt_one_minus_alphas_bar=(1.0 - ab).sqrt(),
        sqrt_recip_alphas_bar=(1.0 / ab).sqrt(), sqrt_recip_m1_alphas_bar=(1.0 / ab - 1.0).sqrt(),
        posterior_var=post_var, posterior_logvar_clipped=post_logvar,
        posterior_mean_coef1=coef1, posterior_mean_coef2=coef2,
        subsequence=torch.as_tensor(subsequence), model_var_type=model_var_type,
    )
    if model_var_type == "fixed-large":
        T["fixed_model_var"] = b
        T["fixed_model_logvar"] = torch.log(torch.cat([post_var[1:2], b[1:]]).clip(min=1e-20))
    else:
        T["fixed_model_var"] = post_var
        T["fixed_model_logvar"] = post_logvar
    return T


def _gather(arr, t, x):
    """diffusion.py:75-84 — cast to x.dtype BEFORE the gather, then broadcast shape."""
    return arr.to(x.dtype).gather(0, t).reshape((-1,) + (1,) * (x.ndim - 1))


def q_sample(T, x_0, t, noise):
    """diffusion.py:92-97."""
    return _gather(T["sqrt_alphas_bar"], t, x_0) * x_0 + _gather(T["sqrt_one_minus_alphas_bar"], t, x_0) * noise


def mse_eps_loss(eps_hat, noise):
    """diffusion.py:236-239 with functions.py:99-101 (per-sample mean over C,H,W)."""
    return ((noise - eps_hat) ** 2).flatten(1).mean(1)


def p_step_from_eps(T, x_t, t, eps_hat, z, clip_denoised=True):
    """diffusion.py:107-158 for model_mean_type='eps', fixed variance.

    Returns (x_{t-1}, pred_x0).  z is the per-step standard normal (drawn even at t=0, then masked).
    """
    x0 = _gather(T["sqrt_recip_alphas_bar"], t, x_t) * x_t - _gather(T["sqrt_recip_m1_alphas_bar"], t, x_t) * eps_hat
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0)
    mean = _gather(T["posterior_mean_coef1"], t, x_t) * x0 + _gather(T["posterior_mean_coef2"], t, x_t) * x_t
    logvar = _gather(T["fixed_model_logvar"], t, x_t)
    mask = (t > 0).reshape((-1,) + (1,) * (x_t.ndim - 1)).to(x_t)
    return mean + mask * torch.exp(0.5 * logvar) * z, x0


def sample_loop(T, denoise_fn, x_T, zs, timestep_map=None):
    """diffusion.py:160-174 / ddim.py:96-113 with injected noise: x_T, then zs[i] for loop step i
    (i = 0 is the first step, t = S-1).  ``timestep_map`` is DDIM's subsequence."""
    S = len(T["posterior_var"])
    x = x_T
    B = x.shape[0]
    for i, ti in enumerate(range(S - 1, -1, -1)):
        t = torch.full((B,), ti, dtype=torch.int64)
        t_model = t if timestep_map is None else timestep_map.gather(0, t)
        eps_hat = denoise_fn(x, t_model)
        x, _ = p_step_from_eps(T, x, t, eps_hat, zs[i])
    return x


# --------------------------------------------------------------------------- variational bound in bits per dimension (loss_type "kl")

def normal_kl(mean1, logvar1, mean2, logvar2):
    """functions.py:30-36."""
    d = logvar1 - logvar2
    return 0.5 * ((-1.0 - d) + (mean1 - mean2) ** 2 * torch.exp(-logvar2) + torch.exp(d))


def approx_std_normal_cdf(x):
    """functions.py:39-46 (Page 1977)."""
    return 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def discretized_gaussian_loglik(x, means, log_scale, precision=1.0 / 255, cutoff=(-0.999, 0.999), tol=1e-12):
    """functions.py:49-65."""
    xc = x - means
    inv_stdv = torch.exp(-log_scale)
    cdf_upper = torch.where(x > cutoff[1], torch.ones_like(x), approx_std_normal_cdf(inv_stdv * (xc + precision)))
    cdf_lower = torch.where(x < cutoff[0], torch.zeros_like(x), approx_std_normal_cdf(inv_stdv * (xc - precision)))
    return torch.log(torch.clamp(cdf_upper - cdf_lower - tol, min=0).add(tol))


def model_mean_from_output(T, mean_type, x_t, t, out, clip_denoised):
    """diffusion.py:120-138 + :140-148: (model_mean, pred_x_0) for model_mean_type in {"eps", "x_0", "mean"}."""
    c1, c2 = _gather(T["posterior_mean_coef1"], t, x_t), _gather(T["posterior_mean_coef2"], t, x_t)
    clip = (lambda v: v.clamp(-1.0, 1.0)) if clip_denoised else (lambda v: v)
    if mean_type == "mean":
        return out, clip(out / c1 - c2 / c1 * x_t)
    if mean_type == "x_0":
        x0 = clip(out)
    elif mean_type == "eps":
        x0 = clip(_gather(T["sqrt_recip_alphas_bar"], t, x_t) * x_t - _gather(T["sqrt_recip_m1_alphas_bar"], t, x_t) * out)
    else:
        raise NotImplementedError(mean_type)
    return c1 * x0 + c2 * x_t, x0


def loss_term_bpd(T, mean_type, x_0, x_t, t, out, clip_denoised):
    """diffusion.py:203-215 `_loss_term_bpd` as a function of the network output ``out``: (per-sample L_t in bits/dim, pred_x_0)."""
    true_mean = _gather(T["posterior_mean_coef1"], t, x_0) * x_0 + _gather(T["posterior_mean_coef2"], t, x_0) * x_t      # :99-105
    true_logvar = _gather(T["posterior_logvar_clipped"], t, x_0)
    model_mean, pred = model_mean_from_output(T, mean_type, x_t, t, out, clip_denoised)
    model_logvar = _gather(T["fixed_model_logvar"], t, x_t)
    kl = normal_kl(true_mean, true_logvar, model_mean, model_logvar).flatten(1).mean(1) / math.log(2.0)
    nll = (-discretized_gaussian_loglik(x_0, model_mean, log_scale=0.5 * model_logvar)).flatten(1).mean(1) / math.log(2.0)
    return torch.where(t > 0, kl, nll), pred
