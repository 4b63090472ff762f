"""Config / seeding glue with the reference's names (``ddpm_torch/utils/__init__.py:39-59,96-101``)."""
import random

import numpy as np
import torch

__all__ = ["seed_all", "get_param", "ConfigDict"]


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
