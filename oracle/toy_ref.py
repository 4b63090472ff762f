"""Oracle (test infrastructure): BASELINE config 1 — the toy 2-D pipeline's MLP denoiser.

Functional restatement of ddpm_torch/toy/toy_model.py:17-62 (Decoder / TemporalLayer).  The
reference repeats ONE TemporalLayer object (toy_model.py:47-48), so the state dict holds a
single set of ``temp_fc.0.*`` tensors shared by every layer; keys ``temp_fc.{i}.*`` alias it.
The toy diffusion (ddpm_torch/toy/diffusion.py:7-64) differs from the image one only in not
clipping x0 — use ``diffusion_ref.p_step_from_eps(..., clip_denoised=False)``.
Not product code.
"""
import torch
import torch.nn.functional as F

from .unet_ref import timestep_embedding


def _lrelu(x):
    return F.leaky_relu(x, 0.02)                                   # toy_model.py:14


def _temporal(sd, p, x, t_emb, d):
    """toy_model.py:32-37 (skip is Identity: in == out)."""
    h = F.linear(_lrelu(F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])), sd[p + "fc1.weight"])
    h = h + F.linear(t_emb, sd[p + "enc.weight"], sd[p + "enc.bias"])
    h = F.linear(_lrelu(F.layer_norm(h, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])), sd[p + "fc2.weight"])
    return h + x


def decoder_forward(sd, x, t, mid_features, num_layers):
    """toy_model.py:56-62."""
    d = mid_features
    t_emb = timestep_embedding(t, d)
    t_emb = _lrelu(F.linear(t_emb, sd["t_proj.0.weight"], sd["t_proj.0.bias"]))
    h = F.linear(x, sd["in_fc.weight"])
    for _ in range(num_layers):
        h = _temporal(sd, "temp_fc.0.", h, t_emb, d)             # shared weights
    h = F.layer_norm(h, (d,), sd["out_norm.weight"], sd["out_norm.bias"])
    return F.linear(h, sd["out_fc.weight"], sd["out_fc.bias"])
