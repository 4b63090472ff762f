"""Streaming activation statistics and the Fréchet distance (``ddpm_torch/metrics/fid_score.py:78-147,160-184,264-316`` of
tqch/ddpm-torch, which wraps the pytorch-fid formulas).  The feature network is pluggable — see the package docstring."""
import os

import numpy as np
import torch

__all__ = ["InceptionStatistics", "get_precomputed", "calc_fd"]

# what the reference would download (kept as a hint for whoever provisions the files; nothing here opens a socket)
PRECOMPUTED_URLS = {
    "celeba": "https://github.com/tqch/VAEGAN/releases/download/precomputed_statistics_celeba/fid_stats_celeba_148x148.npz",
    "lsun_bedroom": "http://bioinf.jku.at/research/ttur/ttur_stats/fid_stats_lsun_train.npz",
    "cifar10": "http://bioinf.jku.at/research/ttur/ttur_stats/fid_stats_cifar10_train.npz",
    "svhn": "http://bioinf.jku.at/research/ttur/ttur_stats/fid_stats_svhn_train.npz",
    "imagenet_train": "http://bioinf.jku.at/research/ttur/ttur_stats/fid_stats_imagenet_train.npz",
    "imagenet_valid": "http://bioinf.jku.at/research/ttur/ttur_stats/fid_stats_imagenet_valid.npz",
}


def _load_extractor(device):
    path = os.environ.get("DDPM_TORCH_AMD_INCEPTION")
    if not path:
        raise RuntimeError(
            "FID needs a feature network: pass feature_extractor= (callable images -> [N, D] activations), or point DDPM_TORCH_AMD_INCEPTION at a "
            "TorchScript export of the reference's Inception-v3 pool3 head.  The pretrained weights cannot be downloaded in this environment, "
            "and no other network is substituted silently.")
    net = torch.jit.load(path, map_location=device if not isinstance(device, list) else device[0]).eval()
    return lambda x: net(x)


class InceptionStatistics:
    """Running mean and covariance of feature activations over batches — ``__call__(images)`` then ``get_statistics()``.

    The merge is the pairwise update the reference applies per batch (fid_score.py:120-132): with n samples seen and a batch of m,
    a = m / (n + m), d = mean_b - mean:   mean += a d;   cov += a (cov_b - cov) + a (1 - a) d d^T   (biased covariances, fp64);
    ``get_statistics`` returns the unbiased covariance (x n / (n - 1)).  Any split of the same samples into batches gives the same
    statistics to rounding."""

    def __init__(self, input_transform=None, activation_dim=2048, device=torch.device("cpu"), feature_extractor=None):
        self.input_transform = input_transform
        self.activation_dim = int(activation_dim)
        self.device = device
        self.model = feature_extractor if feature_extractor is not None else _load_extractor(device)
        self.running_mean = np.zeros((self.activation_dim,), dtype=np.float64)
        self.running_var = np.zeros((self.activation_dim, self.activation_dim), dtype=np.float64)
        self.count = 0

    def features(self, x):
        if self.input_transform is not None:
            x = self.input_transform(x)
        with torch.inference_mode():
            act = self.model(x)
        act = torch.as_tensor(act)
        if act.ndim == 4:                                   # un-pooled maps: global average (fid_score.py:115-116)
            act = act.mean(dim=(2, 3))
        if act.ndim != 2 or act.shape[1] != self.activation_dim:
            raise ValueError(f"feature extractor returned {tuple(act.shape)}, expected [N, {self.activation_dim}]")
        return act.detach().to("cpu", torch.float64).numpy()

    def update(self, act):
        """Merge a batch of activations [m, D] (numpy / tensor) into the running statistics."""
        act = np.asarray(act, dtype=np.float64)
        m = act.shape[0]
        if m == 0:
            return
        mean_b = act.mean(axis=0)
        centred = act - mean_b
        cov_b = centred.T @ centred / m
        if self.count == 0:
            self.running_mean[...] = mean_b
            self.running_var[...] = cov_b
        else:
            a = m / (self.count + m)
            d = mean_b - self.running_mean
            self.running_mean += a * d
            self.running_var += a * (cov_b - self.running_var) + (a * (1.0 - a)) * np.outer(d, d)
        self.count += m

    def __call__(self, x):
        self.update(self.features(x))

    forward = __call__

    def get_statistics(self):
        if self.count <= 1:
            raise AssertionError("Count must be greater than 1!")
        return self.running_mean.copy(), self.running_var * (self.count / (self.count - 1))

    def reset(self):
        self.running_mean.fill(0)
        self.running_var.fill(0)
        self.count = 0


def get_precomputed(dataset, download_dir="precomputed"):
    """(mu, sigma) of the dataset's reference activations from ``<download_dir>/fid_stats_<...>.npz`` (keys "mu", "sigma") — the file the
    reference downloads on first use (fid_score.py:160-183).  No network here: a missing file is an error that names the URL."""
    url = PRECOMPUTED_URLS.get(dataset, f"fid_stats_{dataset}.npz")
    # the published file's name first, then the name eval.py saves statistics computed from the raw data under (eval.py:96)
    names = [os.path.basename(url), f"fid_stats_{dataset}.npz"]
    path = next((p for p in (os.path.join(download_dir or ".", n) for n in names) if os.path.exists(p)), None)
    if path is None:
        raise FileNotFoundError(f"{os.path.join(download_dir or '.', names[0])} not found: place the precomputed statistics of '{dataset}' there "
                                f"({url}); this build does not download")
    with np.load(path) as data:
        return data["mu"], data["sigma"]


def _trace_sqrt_product(s1, s2):
    """tr sqrt(s1 s2) for symmetric positive semi-definite s1, s2: the eigenvalues of s1 s2 are those of the symmetric
    r s2 r with r = s1^(1/2), so two symmetric eigen-decompositions replace the general matrix square root (no complex arithmetic)."""
    w, v = np.linalg.eigh((s1 + s1.T) / 2)
    r = (v * np.sqrt(np.clip(w, 0, None))) @ v.T
    mid = r @ s2 @ r
    ev = np.linalg.eigvalsh((mid + mid.T) / 2)
    return float(np.sqrt(np.clip(ev, 0, None)).sum())


def calc_fd(mean1, var1, mean2, var2, eps=1e-6):
    """Fréchet distance between N(mean1, var1) and N(mean2, var2): |mean1 - mean2|^2 + tr(var1 + var2 - 2 sqrt(var1 var2))
    (fid_score.py:264-316).  A non-finite product term is retried with eps on both diagonals, as the reference does."""
    mean1, mean2 = np.atleast_1d(np.asarray(mean1, dtype=np.float64)), np.atleast_1d(np.asarray(mean2, dtype=np.float64))
    var1, var2 = np.atleast_2d(np.asarray(var1, dtype=np.float64)), np.atleast_2d(np.asarray(var2, dtype=np.float64))
    if mean1.shape != mean2.shape or var1.shape != var2.shape:
        raise AssertionError("the two sets of statistics have different dimensions")
    diff = mean1 - mean2
    tr = _trace_sqrt_product(var1, var2)
    if not np.isfinite(tr):
        off = np.eye(var1.shape[0]) * eps
        tr = _trace_sqrt_product(var1 + off, var2 + off)
    return float(diff @ diff + np.trace(var1) + np.trace(var2) - 2.0 * tr)
