"""Oracle (test infrastructure): import the upstream reference in the BUILD CONTAINER only.

``/root/reference`` does not exist on the GPU box; nothing under ``-m gpu``, ``smoke()`` or
``bench.py`` may call this.  Plain ``import ddpm_torch`` fails there at ``datasets.py`` (no
torchvision), so the hot-path modules are loaded through a stub package.
Used by ``tests/golden/make_golden.py`` and by ``tests/test_oracle_vs_reference.py`` (skipped
when the reference is absent).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DDPM_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ddpm_torch"))


def load():
    """Returns a namespace with the reference's UNet, GaussianDiffusion, DDIM, Trainer, EMA, ..."""
    if not available():
        raise RuntimeError("reference not mounted at " + REFERENCE_ROOT)
    pkg_dir = os.path.join(REFERENCE_ROOT, "ddpm_torch")
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "ddim" or k.startswith(("ddpm_torch", "torchvision"))}
    for k in saved:
        sys.modules.pop(k, None)
    try:
        def stub(name, path=None):
            m = types.ModuleType(name)
            if path:
                m.__path__ = [path]
            sys.modules[name] = m
            return m

        root = stub("ddpm_torch", pkg_dir)
        stub("ddpm_torch.models", os.path.join(pkg_dir, "models"))
        stub("ddpm_torch.toy", os.path.join(pkg_dir, "toy"))
        uts = stub("ddpm_torch.utils", os.path.join(pkg_dir, "utils"))
        uts.save_scatterplot = lambda *a, **k: None
        tv = stub("torchvision"); tvu = stub("torchvision.utils"); tv.utils = tvu
        tvu.save_image = lambda *a, **k: None
        ns = types.SimpleNamespace()
        ns.functions = importlib.import_module("ddpm_torch.functions")
        ns.modules = importlib.import_module("ddpm_torch.modules")
        ns.unet = importlib.import_module("ddpm_torch.models.unet")
        ns.diffusion = importlib.import_module("ddpm_torch.diffusion")
        root.GaussianDiffusion = ns.diffusion.GaussianDiffusion
        root.get_beta_schedule = ns.diffusion.get_beta_schedule
        spec = importlib.util.spec_from_file_location("ddim", os.path.join(REFERENCE_ROOT, "ddim.py"))
        ns.ddim = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ns.ddim)
        ns.train = importlib.import_module("ddpm_torch.utils.train")
        ns.toy_model = importlib.import_module("ddpm_torch.toy.toy_model")
        ns.toy_diffusion = importlib.import_module("ddpm_torch.toy.diffusion")
        ns.toy_utils = importlib.import_module("ddpm_torch.toy.toy_utils")
        ns.UNet = ns.unet.UNet
        ns.GaussianDiffusion = ns.diffusion.GaussianDiffusion
        ns.get_beta_schedule = ns.diffusion.get_beta_schedule
        ns.DDIM = ns.ddim.DDIM
        ns.get_selection_schedule = ns.ddim.get_selection_schedule
        ns.Trainer = ns.train.Trainer
        ns.EMA = ns.train.EMA
        return ns
    finally:
        for k in [k for k in sys.modules if k == "ddim" or k.startswith(("ddpm_torch", "torchvision"))]:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def load_metrics():
    """The reference's evaluation arithmetic (``ddpm_torch/metrics/fid_score.py``, ``precision_recall.py``) for fixture G15.  Those modules
    import torchvision and the Inception wrapper at module level; both are stubbed (nothing of them is called: the fixture drives the
    statistics merge, the Fréchet distance and the manifold functions with synthetic features)."""
    if not available():
        raise RuntimeError("reference not mounted at " + REFERENCE_ROOT)
    pkg_dir = os.path.join(REFERENCE_ROOT, "ddpm_torch")
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.startswith(("ddpm_torch", "torchvision"))}
    for k in saved:
        sys.modules.pop(k, None)
    try:
        def stub(name, path=None):
            m = types.ModuleType(name)
            if path:
                m.__path__ = [path]
            sys.modules[name] = m
            return m

        stub("ddpm_torch", pkg_dir)
        stub("ddpm_torch.metrics", os.path.join(pkg_dir, "metrics"))
        inc = stub("ddpm_torch.metrics.inception")
        inc.InceptionV3 = type("InceptionV3", (), {"BLOCK_INDEX_BY_DIM": {2048: 3}})
        tv = stub("torchvision")
        tf = stub("torchvision.transforms")
        tv.transforms = tf
        for name in ("Compose", "Resize", "Normalize", "ToTensor"):
            setattr(tf, name, type(name, (), {"__init__": lambda self, *a, **k: None}))
        tf.InterpolationMode = types.SimpleNamespace(BILINEAR="bilinear")
        ns = types.SimpleNamespace()
        ns.fid_score = importlib.import_module("ddpm_torch.metrics.fid_score")
        ns.precision_recall = importlib.import_module("ddpm_torch.metrics.precision_recall")
        return ns
    finally:
        for k in [k for k in sys.modules if k.startswith(("ddpm_torch", "torchvision"))]:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
