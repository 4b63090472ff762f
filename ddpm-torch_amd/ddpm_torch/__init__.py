"""MI355X-native drop-in for the hot path of tqch/ddpm-torch.

Put this directory's parent (``ddpm-torch_amd/``) on ``sys.path`` in place of the reference checkout:
``import ddpm_torch`` / ``import ddim`` then resolve here and ``UNet`` / ``GaussianDiffusion`` / ``DDIM`` /
``Trainer`` / ``EMA`` keep the reference's constructor and call contracts (``ddpm_torch/__init__.py:8-22``).
Everything numeric runs in hand-written gfx950 kernels behind ``csrc/libddpm_hip.so``; there is no CPU fallback.
"""
from .datasets import DATASET_DICT, DATASET_INFO, get_dataloader
from .diffusion import GaussianDiffusion, get_beta_schedule
from .metrics import Evaluator
from .models import UNet
from .utils import ConfigDict, get_param, seed_all
from .utils.train import EMA, DummyScheduler, ModelWrapper, Trainer

# the names the reference's CLIs pull in with ``from ddpm_torch import *`` (ddpm_torch/__init__.py:8-22)
__all__ = ["GaussianDiffusion", "get_beta_schedule", "UNet", "seed_all", "get_param", "ConfigDict", "Trainer", "EMA",
           "DummyScheduler", "ModelWrapper", "DATASET_INFO", "DATASET_DICT", "get_dataloader", "Evaluator"]
