"""Host-side functional helpers mirroring ``ddpm_torch/functions.py`` of the reference for the hot path:
``get_timestep_embedding`` (functions.py:10-26) and ``flat_mean`` (functions.py:99-101).  Device work goes
through the C-ABI; CPU tensors raise (no fallback)."""
import math

import torch

from . import _hip

_freq_cache = {}


def _freqs(embed_dim, device):
    key = (embed_dim, device)
    if key not in _freq_cache:
        half = embed_dim // 2
        rate = math.log(10000) / (half - 1)
        _freq_cache[key] = torch.exp(-torch.arange(half, dtype=torch.float32) * rate).to(device)
    return _freq_cache[key]


def get_timestep_embedding(timesteps, embed_dim, dtype=torch.float32):
    """[B] int64 -> [B, embed_dim] fp32: cat(sin(t f_i), cos(t f_i)), f_i = exp(-i ln(1e4)/(half-1)); zero-padded if odd."""
    if dtype != torch.float32:
        raise TypeError("the timestep embedding is fp32-only (as in the reference, functions.py:11)")
    _hip.require_cuda(timesteps)
    t = timesteps.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty((t.numel(), embed_dim), dtype=torch.float32, device=t.device)
    _hip.call("ddpm_timestep_embedding", t.data_ptr(), _freqs(embed_dim, t.device).data_ptr(), out.data_ptr(), t.numel(), embed_dim, _hip.stream())
    return out


def flat_mean(x, start_dim=1):
    return torch.mean(x, dim=list(range(start_dim, x.ndim)))
