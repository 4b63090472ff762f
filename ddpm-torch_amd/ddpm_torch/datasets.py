"""Data path behind the reference's names (``ddpm_torch/datasets.py:13-266`` of tqch/ddpm-torch).

The reference builds its loaders on torchvision dataset classes; torchvision is not part of this image and image decoding is outside
the accelerated hot path, so the four datasets are read from their ON-DISK FORMATS directly (numpy + PIL):

* ``MNIST``      ``<root>/MNIST/raw/{train,t10k}-images-idx3-ubyte[.gz]``, 28×28 → 32×32 bilinear (datasets.py:28-45);
* ``CIFAR10``    ``<root>/cifar-10-batches-py/{data_batch_1..5,test_batch}`` (pickled dicts, opened with a whitelist unpickler),
                 random horizontal flip in training form (:48-66);
* ``CelebA``     ``<root>/celeba/list_eval_partition.txt`` + ``img_align_celeba/``, crop (top 40, left 15, 148×148) → 64×64 bilinear,
                 flip (:69-153); splits train / valid / test / all;
* ``CelebA_HQ``  ``<root>/celeba_hq/img_celeba_hq/*.png`` in ProGAN's legacy order (numeric sort, ``RandomState(123).shuffle``), flip
                 (:156-198).

Every class yields one image per item and no label: a float ``[C, H, W]`` tensor in [-1, 1] (``ToTensor`` + ``Normalize(0.5, 0.5)``),
or — ``transform=None`` / ``raw=True``, what the evaluation code reads — the uint8 ``[C, H, W]`` tensor.  ``DATASET_INFO`` carries the
class constants (resolution, channels, split sizes) the CLIs size their models and sample grids from.

Two more sources exist for boxes without the datasets (this build container and the GPU box have neither the files nor a network):
a **tensor file** ``<root>/<dataset>[_<split>].pt`` (uint8 or [-1, 1] float ``[N, C, H, W]``; tried FIRST, it is also the fastest way to
feed a 10 ms training step) and, when ``DDPM_TORCH_AMD_SYNTHETIC_DATA=N`` is set, N seeded uniform images (smoke / dry runs).

``get_dataloader(dataset, batch_size, split, ...) -> (loader, sampler)`` keeps the reference's contract (:225-266): per-rank batch =
``batch_size // WORLD_SIZE`` when ``distributed``, ``DistributedSampler`` with per-epoch reshuffling, no shuffling for raw / test reads.
"""
import gzip
import io
import os
import pickle
import struct

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

__all__ = ["DATASET_INFO", "DATASET_DICT", "get_dataloader", "TensorImages", "MNIST", "CIFAR10", "CelebA", "CelebA_HQ", "ImageFolder"]

ROOT = os.path.expanduser("~/datasets")
DATASET_DICT, DATASET_INFO = {}, {}


def _register(cls):
    """``DATASET_DICT[name] = cls`` and the class's plain constants into ``DATASET_INFO[name]`` (datasets.py:17-26)."""
    name = cls.__name__.lower()
    DATASET_DICT[name] = cls
    DATASET_INFO[name] = {k: v for k, v in vars(cls).items() if not k.startswith("_") and not callable(v) and not isinstance(v, (staticmethod, classmethod, property))}
    return cls


def _pil():
    try:
        from PIL import Image
    except ImportError as e:                                                         # pragma: no cover
        raise RuntimeError("reading image files needs Pillow; use a tensor file (<root>/<dataset>.pt) instead") from e
    return Image


def _to_unit(x):
    """uint8 [C, H, W] -> float in [-1, 1] exactly as ToTensor + Normalize(0.5, 0.5) round: (x / 255 - 0.5) / 0.5."""
    return x.to(torch.float32).div_(255).sub_(0.5).div_(0.5)


class _Images(Dataset):
    """Common part: ``_load(i)`` -> uint8 [C, H, W]; ``transform=None`` (the reference's raw form) returns it, the default training form
    flips (where the reference's transform does) and normalises.  ``transform`` may also be any callable on the uint8 tensor."""
    flip = False

    def __init__(self, root=ROOT, split="train", transform="train"):
        self.root, self.split, self.transform = os.path.expanduser(root), split, transform

    def _load(self, index):
        raise NotImplementedError

    def __getitem__(self, index):
        x = self._load(index)
        if self.transform is None:
            return x
        if callable(self.transform):
            return self.transform(x)
        if self.flip and torch.rand(1) < 0.5:                  # RandomHorizontalFlip: one uniform draw per image
            x = x.flip(-1)
        return _to_unit(x)


def _read_idx_images(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        magic, n, h, w = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise ValueError(f"{path}: not an idx3-ubyte image file (magic {magic})")
        data = np.frombuffer(f.read(n * h * w), dtype=np.uint8)
    if data.size != n * h * w:
        raise ValueError(f"{path}: truncated ({data.size} of {n * h * w} bytes)")
    return data.reshape(n, h, w)


@_register
class MNIST(_Images):
    resolution = (32, 32)
    channels = 1
    train_size = 60000
    test_size = 10000

    def __init__(self, root=ROOT, split="train", transform="train"):
        super().__init__(root, split, transform)
        stem = "t10k" if split == "test" else "train"
        base = os.path.join(self.root, "MNIST", "raw", f"{stem}-images-idx3-ubyte")
        path = next((p for p in (base, base + ".gz") if os.path.exists(p)), None)
        if path is None:
            raise FileNotFoundError(base)
        self.data = _read_idx_images(path)

    def __len__(self):
        return len(self.data)

    def _load(self, index):
        Image = _pil()
        im = Image.fromarray(self.data[index]).resize(self.resolution[::-1], Image.BILINEAR)
        return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).unsqueeze(0)


class _ArraysOnly(pickle.Unpickler):
    """The CIFAR batches are pickled dicts of lists, strings and one numpy array: nothing else is allowed to be constructed."""
    ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
               ("_codecs", "encode")}

    def find_class(self, module, name):
        if (module, name) not in self.ALLOWED:
            raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a dataset file")
        return super().find_class(module, name)


@_register
class CIFAR10(_Images):
    resolution = (32, 32)
    channels = 3
    train_size = 50000
    test_size = 10000
    flip = True

    def __init__(self, root=ROOT, split="train", transform="train"):
        super().__init__(root, split, transform)
        folder = os.path.join(self.root, "cifar-10-batches-py")
        names = ["test_batch"] if split == "test" else [f"data_batch_{i}" for i in range(1, 6)]
        parts = []
        for fn in names:
            path = os.path.join(folder, fn)
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            with open(path, "rb") as f:
                entry = _ArraysOnly(io.BytesIO(f.read()), encoding="latin1").load()
            parts.append(np.asarray(entry["data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
        self.data = np.concatenate(parts)

    def __len__(self):
        return len(self.data)

    def _load(self, index):
        return torch.from_numpy(self.data[index].copy())


def _open_rgb(path):
    with _pil().open(path) as im:
        return im.convert("RGB") if im.mode != "RGB" else im.copy()


def _chw(im):
    return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous()


@_register
class CelebA(_Images):
    base_folder = "celeba"
    resolution = (64, 64)
    channels = 3
    all_size = 202599
    train_size = 162770
    val_size = 19867
    test_size = 19962
    flip = True

    def __init__(self, root=ROOT, split="train", transform="train"):
        super().__init__(root, split, transform)
        want = {"train": 0, "valid": 1, "test": 2, "all": None}[split.lower()]
        part = os.path.join(self.root, self.base_folder, "list_eval_partition.txt")
        if not os.path.exists(part):
            raise FileNotFoundError(part)
        self.filename = []
        with open(part) as f:
            for line in f:
                fields = line.split()
                if len(fields) >= 2 and (want is None or int(fields[1]) == want):
                    self.filename.append(fields[0])

    def __len__(self):
        return len(self.filename)

    def _load(self, index):
        Image = _pil()
        im = _open_rgb(os.path.join(self.root, self.base_folder, "img_align_celeba", self.filename[index]))
        im = im.crop((15, 40, 15 + 148, 40 + 148)).resize(self.resolution[::-1], Image.BILINEAR)      # (left, top, right, bottom)
        return _chw(im)


@_register
class CelebA_HQ(_Images):
    base_folder = "celeba_hq"
    resolution = (256, 256)
    channels = 3
    all_size = 30000
    train_size = 24000                       # (this package's CLI reports progress against it; the reference trains on all 30000)
    flip = True

    def __init__(self, root=ROOT, split="train", transform="train"):
        super().__init__(root, split, transform)             # split is unused, as in the reference
        folder = os.path.join(self.root, self.base_folder, "img_celeba_hq")
        if not os.path.isdir(folder):
            raise FileNotFoundError(folder)
        self.filename = sorted((fn for fn in os.listdir(folder) if fn.endswith(".png")), key=lambda fn: int(fn[:-4]))
        np.random.RandomState(123).shuffle(self.filename)      # legacy ProGAN order

    def __len__(self):
        return len(self.filename)

    def _load(self, index):
        return _chw(_open_rgb(os.path.join(self.root, self.base_folder, "img_celeba_hq", self.filename[index])))


class ImageFolder(Dataset):
    """Every image file of one folder as a uint8 [C, H, W] tensor — the generated samples ``eval.py`` scores (eval.py:56-70)."""
    EXT = {"jpg", "jpeg", "png", "bmp", "webp", "tiff"}

    def __init__(self, img_dir, transform=None):
        self.img_dir = img_dir
        self.img_list = sorted(fn for fn in os.listdir(img_dir) if fn.rsplit(".", 1)[-1].lower() in self.EXT)
        self.transform = transform

    def __len__(self):
        return len(self.img_list)

    def __getitem__(self, index):
        with _pil().open(os.path.join(self.img_dir, self.img_list[index])) as im:
            x = np.asarray(im, dtype=np.uint8)
        x = torch.from_numpy(x.copy())
        x = x.unsqueeze(0) if x.ndim == 2 else x.permute(2, 0, 1).contiguous()
        return x if self.transform is None else self.transform(x)


class TensorImages(Dataset):
    """Images held in one tensor; ``__getitem__`` returns a float image in [-1, 1] (unconditional: no label), or the stored uint8 image
    when ``raw``."""

    def __init__(self, data, flip=False, raw=False):
        assert data.ndim == 4, "expected [N, C, H, W]"
        self.data, self.flip, self.raw = data, flip, raw

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        x = self.data[i]
        if self.raw:
            return x if x.dtype == torch.uint8 else (x * 127.5 + 128).clamp(0, 255).to(torch.uint8)
        if x.dtype == torch.uint8:
            x = x.float().div_(127.5).sub_(1.0)
        if self.flip and torch.rand(()) < 0.5:          # RandomHorizontalFlip of the reference's training transforms
            x = x.flip(-1)
        return x


def _open(name, root, split, raw=False):
    """Tensor file, then the dataset's own files, then (if asked for) synthetic images."""
    info = DATASET_INFO[name]
    shape = (info["channels"],) + tuple(info["resolution"])
    root = os.path.expanduser(root)
    for fn in (f"{name}_{split}.pt", f"{name}.pt"):
        path = os.path.join(root, fn)
        if os.path.exists(path):
            data = torch.load(path, map_location="cpu")
            assert tuple(data.shape[1:]) == shape, f"{path}: images of shape {tuple(data.shape[1:])}, expected {shape}"
            return TensorImages(data, flip=(DATASET_DICT[name].flip and split != "test"), raw=raw)
    try:
        return DATASET_DICT[name](root=root, split=split, transform=None if raw else "train")
    except FileNotFoundError as e:
        missing = e
    n = int(os.environ.get("DDPM_TORCH_AMD_SYNTHETIC_DATA", "0"))
    if n > 0:
        g = torch.Generator().manual_seed(1234)
        return TensorImages(torch.rand((n,) + shape, generator=g) * 2 - 1, raw=raw)
    raise FileNotFoundError(
        f"{name}: neither a tensor file {name}.pt under {root} ([N, C, H, W] uint8 or float) nor the dataset's own files ({missing}).  "
        f"Set DDPM_TORCH_AMD_SYNTHETIC_DATA=<N> for N synthetic images of shape {shape}.")


def get_dataloader(dataset, batch_size, split, val_size=0., random_seed=None, root=ROOT, pin_memory=False, drop_last=False,
                   num_workers=0, distributed=False, raw=False, **kwargs):
    """(DataLoader, sampler) with the reference's batch / shard semantics (datasets.py:225-266)."""
    data = _open(dataset, root, split, raw=raw)
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
    shuffle = sampler is None and split in ("train", "all") and not raw
    loader = DataLoader(data, batch_size=batch_size, shuffle=shuffle, sampler=sampler, drop_last=drop_last,
                        pin_memory=pin_memory and torch.cuda.is_available(), num_workers=num_workers)
    return loader, sampler
