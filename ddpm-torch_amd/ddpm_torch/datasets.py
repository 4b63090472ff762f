"""Data path stand-ins behind the reference's names (``ddpm_torch/datasets.py:13-266`` of tqch/ddpm-torch).

The reference builds its loaders on torchvision datasets (MNIST / CIFAR10 / CelebA / CelebA-HQ folders); torchvision is
not part of this image and the image-decoding data path is outside the accelerated hot path.  What the
CLIs need from this module is kept, with the same call contract:

* ``DATASET_INFO`` / ``DATASET_DICT`` — registry of dataset names with their resolution and channel count;
* ``get_dataloader(dataset, batch_size, split, ...) -> (loader, sampler)`` — per-rank batch = ``batch_size // WORLD_SIZE``
  when ``distributed`` (datasets.py:244-245), ``DistributedSampler`` with per-epoch reshuffling (:262-263), ``drop_last``
  for a static input shape.  Images come from a **tensor file** ``<root>/<dataset>.pt`` (uint8 ``[N, C, H, W]`` or float in
  [-1, 1]; uint8 is mapped to [-1, 1] like ``Normalize(0.5, 0.5)``), or — when no such file exists and
  ``DDPM_TORCH_AMD_SYNTHETIC_DATA=N`` is set — from N seeded uniform images of the dataset's shape (smoke / dry runs).
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

__all__ = ["DATASET_INFO", "DATASET_DICT", "get_dataloader", "TensorImages"]

ROOT = os.path.expanduser("~/datasets")

# shapes the reference's dataset classes declare (datasets.py:29-31,48-50,77-79,163-165)
DATASET_INFO = {
    "mnist": {"resolution": (32, 32), "channels": 1, "train_size": 60000, "test_size": 10000},
    "cifar10": {"resolution": (32, 32), "channels": 3, "train_size": 50000, "test_size": 10000},
    "celeba": {"resolution": (64, 64), "channels": 3, "train_size": 162770},
    "celeba_hq": {"resolution": (256, 256), "channels": 3, "train_size": 24000},
}


class TensorImages(Dataset):
    """Images held in one tensor; ``__getitem__`` returns a float image in [-1, 1] (unconditional: no label)."""

    def __init__(self, data, flip=False):
        assert data.ndim == 4, "expected [N, C, H, W]"
        self.data, self.flip = data, flip

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        x = self.data[i]
        if x.dtype == torch.uint8:
            x = x.float().div_(127.5).sub_(1.0)
        if self.flip and torch.rand(()) < 0.5:          # RandomHorizontalFlip of the reference's training transforms
            x = x.flip(-1)
        return x


def _open(name, root, split):
    info = DATASET_INFO[name]
    shape = (info["channels"],) + tuple(info["resolution"])
    for fn in (f"{name}_{split}.pt", f"{name}.pt"):
        path = os.path.join(os.path.expanduser(root), fn)
        if os.path.exists(path):
            data = torch.load(path, map_location="cpu")
            assert tuple(data.shape[1:]) == shape, f"{path}: images of shape {tuple(data.shape[1:])}, expected {shape}"
            return TensorImages(data, flip=(name == "cifar10" and split != "test"))
    n = int(os.environ.get("DDPM_TORCH_AMD_SYNTHETIC_DATA", "0"))
    if n > 0:
        g = torch.Generator().manual_seed(1234)
        return TensorImages(torch.rand((n,) + shape, generator=g) * 2 - 1)
    raise FileNotFoundError(
        f"no tensor file {name}.pt under {root} (this build reads images from a [N, C, H, W] tensor file; torchvision datasets are "
        f"not available here).  Set DDPM_TORCH_AMD_SYNTHETIC_DATA=<N> for N synthetic images of shape {shape}.")


DATASET_DICT = {name: (lambda root=ROOT, split="train", _n=name, **kw: _open(_n, root, split)) for name in DATASET_INFO}


def get_dataloader(dataset, batch_size, split, val_size=0., random_seed=None, root=ROOT, pin_memory=False, drop_last=False,
                   num_workers=0, distributed=False, raw=False, **kwargs):
    """(DataLoader, sampler) with the reference's batch / shard semantics (datasets.py:225-266)."""
    data = DATASET_DICT[dataset](root=root, split=split)
    if distributed:
        batch_size = batch_size // int(os.environ.get("WORLD_SIZE", "1"))
    if split != "test" and val_size > 0.:
        n = len(data)
        g = torch.Generator().manual_seed(random_seed or 0)
        perm = torch.randperm(n, generator=g)
        n_val = int(n * val_size) if val_size < 1 else int(val_size)
        idx = perm[n_val:] if split == "train" else perm[:n_val]
        data = torch.utils.data.Subset(data, idx.tolist())
    sampler = DistributedSampler(data, shuffle=True, seed=random_seed or 0, drop_last=drop_last) if distributed else None
    loader = DataLoader(data, batch_size=batch_size, shuffle=sampler is None and split != "test", sampler=sampler, drop_last=drop_last,
                        pin_memory=pin_memory and torch.cuda.is_available(), num_workers=num_workers)
    return loader, sampler
