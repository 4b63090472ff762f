"""Evaluation behind the reference's names (``ddpm_torch/metrics/`` of tqch/ddpm-torch): ``Evaluator`` (FID during training,
``metrics/__init__.py:18-53``), streaming activation statistics + Fréchet distance (``fid_score.py:78-147,316``) and the k-NN
precision / recall of Kynkäänniemi et al. (``precision_recall.py:141-206``).

Outside the accelerated hot path: plain torch / numpy, any device.  What the reference downloads — the
pretrained Inception-v3 / VGG-16 feature networks and the datasets' reference statistics — cannot be fetched here, so the FEATURE
EXTRACTOR IS PLUGGABLE: pass ``feature_extractor=`` (a callable ``images -> [N, D]`` or ``[N, D, h, w]`` activations; e.g. the
TorchScript of the reference's network loaded from a local file) and ``target_stats=`` / a local ``precomputed/fid_stats_<dataset>.npz``.
Everything around the network — the streaming mean / covariance merge, the Fréchet distance, the manifold radii and the
precision / recall test, the sample loop of ``Evaluator.eval`` — is implemented and pinned to the reference's own functions by fixture
G15 (tests/test_metrics.py).  Without an extractor the constructors raise: no number is ever produced from something else."""
import math

import torch

from .fid_score import InceptionStatistics, calc_fd, get_precomputed
from .precision_recall import Manifold, ManifoldBuilder, calc_pr, load_manifold

__all__ = ["InceptionStatistics", "get_precomputed", "calc_fd", "ManifoldBuilder", "Manifold", "calc_pr", "load_manifold", "Evaluator"]


class Evaluator:
    """``Evaluator(dataset, diffusion, eval_batch_size, eval_total_size, device)`` + ``eval(sample_fn, is_leader) -> {"fid": ...}``
    as the reference's trainer calls it (utils/train.py:225-226).  Extra keyword arguments (not in the reference):
    ``feature_extractor`` (see the module docstring; default: the TorchScript file named by ``DDPM_TORCH_AMD_INCEPTION``),
    ``target_stats=(mean, cov)`` instead of the precomputed-statistics file, ``precomputed_dir``.

    One deliberate difference: the reference sizes its LAST batch as ``eval_total_size % eval_batch_size`` — zero samples when the batch
    size divides the total (metrics/__init__.py:41-42); here the last batch is whatever is left, a full batch in that case."""

    def __init__(self, dataset, diffusion=None, eval_batch_size=256, eval_total_size=50000, device=torch.device("cpu"),
                 feature_extractor=None, target_stats=None, precomputed_dir="precomputed"):
        self.diffusion = diffusion
        self.istats = InceptionStatistics(device=device, feature_extractor=feature_extractor)
        self.eval_batch_size, self.eval_total_size, self.device = int(eval_batch_size), int(eval_total_size), device
        if self.eval_batch_size <= 0 or self.eval_total_size <= 1:
            raise ValueError("eval_batch_size must be positive and eval_total_size at least 2")
        self.target_mean, self.target_var = target_stats if target_stats is not None else get_precomputed(dataset, precomputed_dir)

    def eval(self, sample_fn, is_leader=True):
        """Every rank takes part in every ``sample_fn`` call (it all-gathers under DDP); only the leader accumulates statistics."""
        if is_leader:
            self.istats.reset()
        fid, left = None, self.eval_total_size
        for _ in range(math.ceil(self.eval_total_size / self.eval_batch_size)):
            n = min(self.eval_batch_size, left)
            left -= n
            x = sample_fn(sample_size=n, diffusion=self.diffusion)
            if is_leader:
                self.istats(x.to(self.device))
        if is_leader:
            gen_mean, gen_var = self.istats.get_statistics()
            fid = calc_fd(gen_mean, gen_var, self.target_mean, self.target_var)
        return {"fid": fid}
