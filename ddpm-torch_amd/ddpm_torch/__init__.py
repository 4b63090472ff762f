"""MI355X-native drop-in for the hot path of tqch/ddpm-torch.

Put this directory's parent (``ddpm-torch_amd/``) on ``sys.path`` in place of the reference checkout:
``import ddpm_torch`` / ``import ddim`` then resolve here and ``UNet`` / ``GaussianDiffusion`` / ``DDIM`` /
``Trainer`` / ``EMA`` keep the reference's constructor and call contracts (``ddpm_torch/__init__.py:8-22``).
Everything numeric runs in hand-written gfx950 kernels behind ``csrc/libddpm_hip.so``; there is no CPU fallback.
"""
from .diffusion import GaussianDiffusion, get_beta_schedule
from .models import UNet
from .utils import ConfigDict, get_param, seed_all
from .utils.train import EMA, DummyScheduler, ModelWrapper, Trainer

# Shape metadata the reference's CLIs read from its dataset registry (ddpm_torch/datasets.py:49-50,78-79,164-165).
DATASET_INFO = {
    "mnist": {"resolution": (32, 32), "channels": 1},
    "cifar10": {"resolution": (32, 32), "channels": 3},
    "celeba": {"resolution": (64, 64), "channels": 3},
    "celeba_hq": {"resolution": (256, 256), "channels": 3},
}

__all__ = ["GaussianDiffusion", "get_beta_schedule", "UNet", "seed_all", "get_param", "ConfigDict", "Trainer", "EMA",
           "DummyScheduler", "ModelWrapper", "DATASET_INFO"]
