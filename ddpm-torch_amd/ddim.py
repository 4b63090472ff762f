"""DDIM sampler (Song et al. 2020) with the reference's module-level contract (``ddim.py:30-120`` of
tqch/ddpm-torch): ``get_selection_schedule`` and ``DDIM(GaussianDiffusion)`` incl. ``from_ddpm``.

Only the tables differ from the ancestral sampler; the loop and the fused device step are inherited from
``ddpm_torch.GaussianDiffusion``.  The model is evaluated at ``subsequence[t]`` (ddim.py:101) via a device gather.
"""
import math

import torch

import ddpm_torch
from ddpm_torch import _hip

__all__ = ["get_selection_schedule", "DDIM"]


def get_selection_schedule(schedule, size, timesteps):
    """Sub-sequence of original timesteps (ddim.py:30-44).  NB 'quadratic' can repeat an entry (size 50 -> two zeros)."""
    assert schedule in {"linear", "quadratic"}
    if schedule == "linear":
        return torch.arange(0, timesteps, timesteps // size)
    return torch.pow(torch.linspace(0, math.sqrt(timesteps * 0.8), size), 2).round().to(torch.int64)


class DDIM(ddpm_torch.GaussianDiffusion):
    def __init__(self, betas, model_mean_type, model_var_type, loss_type, eta, subsequence):
        super().__init__(betas, model_mean_type, model_var_type, loss_type)
        self.eta = eta
        eta2 = eta ** 2
        if eta2 != 1.0 and model_var_type != "fixed-small":
            self.model_var_type = "fixed-small"          # silent coercion, as ddim.py:53-59
        ab = self.alphas_bar[subsequence]
        ab_prev = torch.cat([torch.ones(1, dtype=torch.float64), ab[:-1]], dim=0)
        self.alphas = ab / ab_prev
        self.betas = 1.0 - self.alphas
        self.alphas_bar_prev = self._build_tables(self.betas, ab, eta2=eta2)
        self.sqrt_alphas_bar_prev = torch.sqrt(self.alphas_bar_prev)
        self.subsequence = torch.as_tensor(subsequence)
        self._sub_dev = {}

    def _num_steps(self):
        return len(self.subsequence)

    def _model_t(self, t):
        dev = t.device
        if dev not in self._sub_dev:
            self._sub_dev[dev] = self.subsequence.to(torch.int64).to(dev).contiguous()
        out = torch.empty_like(t)
        _hip.call("ddpm_gather_i64", t.data_ptr(), self._sub_dev[dev].data_ptr(), out.data_ptr(), t.numel(), _hip.stream())
        return out

    def p_sample_step(self, denoise_fn, x_t, t, clip_denoised=True, return_pred=False, generator=None):
        return super().p_sample_step(lambda x, tt: denoise_fn(x, self._model_t(tt)), x_t, t, clip_denoised, return_pred, generator)

    @torch.inference_mode()
    def p_sample(self, denoise_fn, shape, device=torch.device("cpu"), noise=None, seed=None):
        """ddim.py:96-113: S = len(subsequence) steps, same RNG consumption order as the ancestral loop."""
        return self._sample_loop(denoise_fn, shape, device, noise, seed)

    @classmethod
    def from_ddpm(cls, diffusion, eta, subsequence):
        keys = ("betas", "model_mean_type", "model_var_type", "loss_type")
        return cls(**{k: diffusion.__dict__.get(k, None) for k in keys}, eta=eta, subsequence=subsequence)
