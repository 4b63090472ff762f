"""Improved precision / recall (Kynkäänniemi et al., 2019) as the reference computes it (``ddpm_torch/metrics/precision_recall.py:42-206``
of tqch/ddpm-torch): a sample set is represented by its feature vectors and, per vector, the distance to its k-th nearest neighbour;
precision = share of generated features inside at least one real hypersphere, recall the converse.  The VGG-16 feature network is
pluggable (the reference downloads a TorchScript file): ``ManifoldBuilder(..., extractor=callable)`` or precomputed ``features=``."""
import math
import os
from collections import namedtuple

import numpy as np
import torch

__all__ = ["Manifold", "ManifoldBuilder", "calc_pr", "compute_distance", "to_uint8", "load_manifold"]

Manifold = namedtuple("Manifold", ["features", "kth"])


def _pairwise(a, b):
    """Euclidean distances [len(a), len(b)].  fp16 features (how the manifolds are stored) are widened to fp32 for the product and the
    result is rounded back: torch's half-precision cdist is missing on the host for small blocks and overflows |x|^2 > 65504 in its
    matmul form for large ones."""
    if a.dtype in (torch.float16, torch.bfloat16):
        return torch.cdist(a.float().unsqueeze(0), b.float().unsqueeze(0)).squeeze(0).to(a.dtype)
    return torch.cdist(a.unsqueeze(0), b.unsqueeze(0)).squeeze(0)


def compute_distance(row_features, col_features, row_batch_size, col_batch_size, device):
    """All-pairs distances, computed block by block on ``device`` and gathered on the host (precision_recall.py:42-54)."""
    rows = []
    for r in row_features.split(row_batch_size, dim=0):
        r = r.to(device)
        rows.append(torch.cat([_pairwise(r, c.to(device)).cpu() for c in col_features.split(col_batch_size, dim=0)], dim=1))
    return torch.cat(rows, dim=0)


def to_uint8(x):
    """[-1, 1] floats -> uint8 images the way the reference feeds VGG (precision_recall.py:57-58)."""
    return (x * 127.5 + 128).clamp(0, 255).to(torch.uint8)


class ManifoldBuilder:
    """Features (fp16, as the reference stores them) + k-th-neighbour radii of one sample set.

    Sources, first match wins: ``features`` (a tensor [N, D]); ``model`` with ``sample_x(n) -> images in [-1, 1]``; ``data`` (uint8 tensor /
    array / .npy / .pt path, or a map-style dataset).  More than ``max_sample_size`` items are subsampled with numpy's seeded
    ``choice`` like the reference (precision_recall.py:87-90,109-112)."""

    def __init__(self, data=None, model=None, features=None, extr_batch_size=128, max_sample_size=50000, nhood_size=3,
                 row_batch_size=10000, col_batch_size=10000, random_state=1234, num_workers=0, device=torch.device("cpu"), extractor=None):
        self.op_device = device[0] if isinstance(device, list) else device
        self.nhood_size, self.row_batch_size, self.col_batch_size, self.device = nhood_size, row_batch_size, col_batch_size, device
        if features is None:
            if extractor is None:
                raise RuntimeError("ManifoldBuilder needs features= or extractor= (callable uint8 images -> [N, D] features): the VGG-16 "
                                   "TorchScript the reference downloads is not available offline and nothing is substituted for it")
            feats = []
            with torch.inference_mode():
                for x in self._batches(data, model, extr_batch_size, max_sample_size, random_state, num_workers):
                    feats.append(torch.as_tensor(extractor(x.to(self.op_device))).cpu())
            features = torch.cat(feats, dim=0)
        elif not isinstance(features, torch.Tensor) or features.grad_fn is not None:
            raise AssertionError("features must be a tensor outside any autograd graph")
        self.features = features.to(torch.float16)
        self.kth = self.compute_kth(self.features)

    @staticmethod
    def _batches(data, model, bs, cap, seed, workers):
        if model is not None:
            left = cap
            while left > 0:
                n = min(bs, left)
                left -= n
                yield to_uint8(model.sample_x(n))
            return
        if isinstance(data, str):
            data = np.load(data) if data.endswith(".npy") else torch.load(data, map_location="cpu")
        if isinstance(data, (np.ndarray, torch.Tensor)):
            data = torch.as_tensor(data)
            if data.dtype != torch.uint8:
                raise AssertionError("image tensors must be uint8")
            if data.shape[0] > cap:
                np.random.seed(seed)
                data = data[torch.as_tensor(np.random.choice(data.shape[0], size=cap, replace=False))]
            for i in range(0, data.shape[0], bs):
                yield data[i:i + bs]
            return
        from torch.utils.data import DataLoader, Subset
        if len(data) > cap:
            np.random.seed(seed)
            data = Subset(data, indices=torch.as_tensor(np.random.choice(len(data), size=cap, replace=False)))
        for x in DataLoader(data, batch_size=bs, shuffle=False, num_workers=workers, drop_last=False):
            yield x[0] if isinstance(x, (list, tuple)) else x

    def compute_distance(self, row_features, col_features):
        return compute_distance(row_features, col_features, self.row_batch_size, self.col_batch_size, self.op_device)

    def compute_kth(self, row_features, col_features=None):
        """Distance of every row feature to its ``nhood_size``-th nearest OTHER feature (k + 1 smallest: the feature itself is at 0)."""
        cols = row_features if col_features is None else col_features
        out = []
        for block in row_features.split(self.row_batch_size, dim=0):
            d = self.compute_distance(block, cols).to(torch.float32)
            out.append(d.kthvalue(self.nhood_size + 1, dim=1).values.to(torch.float16))
        return torch.cat(out)

    @property
    def manifold(self):
        return Manifold(features=self.features, kth=self.kth)

    def save(self, fpath):
        os.makedirs(os.path.dirname(fpath) or ".", exist_ok=True)
        torch.save(self.manifold, fpath)


def load_manifold(fpath):
    """A manifold file written by ``ManifoldBuilder.save`` — here or by the reference (the same namedtuple under the same module path) —
    opened with the tensors-only unpickler plus that one class."""
    with torch.serialization.safe_globals([Manifold]):
        features, kth = torch.load(fpath, map_location="cpu", weights_only=True)
    return Manifold(features=features, kth=kth)


def _coverage(probe, support, row_batch_size, col_batch_size, device):
    """Share of probe features that fall inside at least one support hypersphere."""
    hit = []
    for block in probe.features.split(row_batch_size):
        d = compute_distance(block, support.features, row_batch_size, col_batch_size, device)
        hit.append((d <= support.kth.unsqueeze(0)).any(dim=1))
    return torch.cat(hit).to(torch.float32).mean()


def calc_pr(manifold_1, manifold_2, row_batch_size, col_batch_size, device):
    """(precision, recall) with manifold_1 = generated, manifold_2 = real (precision_recall.py:177-206)."""
    return (_coverage(manifold_1, manifold_2, row_batch_size, col_batch_size, device),
            _coverage(manifold_2, manifold_1, row_batch_size, col_batch_size, device))
