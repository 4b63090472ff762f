"""Config / seeding glue with the reference's names (``ddpm_torch/utils/__init__.py:39-59,96-101``)."""
import random

import numpy as np
import torch

__all__ = ["seed_all", "get_param", "ConfigDict", "save_image_grid"]


def seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_param(param, obj_1, obj_2):
    """Look ``param`` up in obj_1 (JSON section), falling back to obj_2 (argparse namespace)."""
    def get(obj, attr):
        return obj[attr] if hasattr(obj, "__getitem__") else getattr(obj, attr)
    try:
        return get(obj_1, param)
    except (KeyError, AttributeError):
        return get(obj_2, param)


class ConfigDict(dict):
    """dict with attribute access; missing keys read as None."""

    def __getattr__(self, name):
        return self.get(name, None)


def save_image_grid(x, path, nrow=8, pad=2):
    """Write a batch of images in [-1, 1] ([N, C, H, W]) as one grid picture — the role torchvision's ``save_image(normalize=True,
    value_range=(-1, 1))`` plays in the reference (utils/train.py:60).  Needs Pillow; without it the tensor is saved next to the
    requested name as ``.pt``."""
    x = x.detach().float().cpu().clamp(-1, 1)
    n, c, h, w = x.shape
    nrow = max(1, min(nrow, n))
    rows = (n + nrow - 1) // nrow
    grid = torch.zeros(c, rows * (h + pad) + pad, nrow * (w + pad) + pad)
    for i in range(n):
        r, q = divmod(i, nrow)
        grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = x[i]
    arr = (grid * 127.5 + 127.5).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    try:
        from PIL import Image
    except ImportError:
        torch.save(x, path.rsplit(".", 1)[0] + ".pt")
        return
    Image.fromarray(arr.squeeze(-1) if c == 1 else arr).save(path)
