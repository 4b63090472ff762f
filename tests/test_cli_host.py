"""CLI counterparts on the CPU with the C ABI emulated: this build's train.py / generate.py end to end on a
tiny configuration, and — in the build container, where /root/reference exists — the reference's OWN, unmodified train.py
running against this package (the drop-in claim of the upper boundary)."""
import glob
import json
import os
import runpy
import sys

import pytest
import torch

from ddpm_torch import _hip
from tests import abi_emulator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ddpm-torch_amd")
pytestmark = pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")

TINY_CFG = {
    "dataset": "cifar10",
    "diffusion": {"timesteps": 40, "beta_start": 1e-4, "beta_end": 0.02, "beta_schedule": "linear", "model_mean_type": "eps",
                  "model_var_type": "fixed-large", "loss_type": "mse"},
    "model": {"in_channels": 3, "hid_channels": 32, "ch_multipliers": [1, 2], "num_res_blocks": 1, "apply_attn": [False, True], "drop_rate": 0.1},
    "train": {"lr": 2e-4, "batch_size": 4, "grad_norm": 1.0, "epochs": 1, "warmup": 10, "use_ema": True, "ema_decay": 0.9999},
}


@pytest.fixture
def emu(monkeypatch):
    return abi_emulator.install(monkeypatch, _hip)


@pytest.fixture
def workdir(tmp_path, monkeypatch):
    cfg = tmp_path / "tiny.json"
    cfg.write_text(json.dumps(TINY_CFG))
    monkeypatch.setenv("DDPM_TORCH_AMD_SYNTHETIC_DATA", "8")
    monkeypatch.chdir(tmp_path)
    return tmp_path, str(cfg)


def _import_cli(name):
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import importlib
    return importlib.import_module(name)


def test_train_then_generate_round_trip(emu, workdir):
    tmp, cfg = workdir
    train = _import_cli("train")
    tr = train.main(["--config-path", cfg, "--train-device", "cpu", "--dry-run", "--num-samples", "2", "--use-ddim", "--num-workers", "0",
                     "--compute", "fp32", "--chkpt-dir", str(tmp / "chk"), "--image-dir", str(tmp / "img")])
    assert tr.ema.num_updates == 0 and int(next(iter(tr.optimizer.state.values()))["step"]) == 1      # exactly one update
    chk = tmp / "chk" / "tiny" / "tiny_1.pt"
    assert chk.exists()
    assert glob.glob(str(tmp / "chk" / "tiny" / "exp_*.info"))                                         # hyper-parameter record
    assert (tmp / "img" / "train" / "tiny" / "1.jpg").exists()                                         # sample grid of the epoch
    rec = json.load(open(glob.glob(str(tmp / "chk" / "tiny" / "exp_*.info"))[0]))
    assert rec["model"]["block_size"] == 1 and rec["train"]["batch_size"] == 4 and rec["diffusion"]["timesteps"] == 40
    # resume: the epoch counter and the EMA state come back
    tr2 = train.main(["--config-path", cfg, "--train-device", "cpu", "--dry-run", "--num-samples", "0", "--resume", "--num-workers", "0",
                      "--compute", "fp32", "--chkpt-path", str(chk), "--chkpt-dir", str(tmp / "chk2"), "--image-dir", str(tmp / "img2")])
    assert tr2.ema.num_updates == 1
    # generate.py: EMA shadow preferred, DDP prefixes stripped, frozen model, PNG files
    saved = torch.load(chk, weights_only=False)
    saved["ema"]["shadow"] = {"module." + k: v for k, v in saved["ema"]["shadow"].items()}
    ddp_chk = tmp / "ddp_style.pt"
    torch.save(saved, ddp_chk)
    gen = _import_cli("generate")
    out = gen.main(["--config-path", cfg, "--device", "cpu", "--chkpt-path", str(ddp_chk), "--use-ddim", "--subseq-size", "4", "--total-size", "3",
                    "--batch-size", "2", "--save-dir", str(tmp / "gen"), "--compute", "fp32", "--seed", "7"])
    files = glob.glob(os.path.join(out, "*.png"))
    assert len(files) == 3
    from PIL import Image
    assert Image.open(files[0]).size == (32, 32)
    # a bare state dict is accepted as well
    torch.save(saved["model"], tmp / "bare.pt")
    out2 = gen.main(["--config-path", cfg, "--device", "cpu", "--chkpt-path", str(tmp / "bare.pt"), "--use-ddim", "--subseq-size", "4",
                     "--total-size", "1", "--batch-size", "2", "--save-dir", str(tmp / "gen2"), "--compute", "fp32"])
    assert len(glob.glob(os.path.join(out2, "*.png"))) == 1


def test_eval_flag_without_a_feature_network_is_refused_loudly(emu, workdir, monkeypatch):
    tmp, cfg = workdir
    train = _import_cli("train")
    monkeypatch.delenv("DDPM_TORCH_AMD_INCEPTION", raising=False)
    with pytest.raises(RuntimeError, match="feature network"):
        train.main(["--config-path", cfg, "--train-device", "cpu", "--dry-run", "--num-samples", "0", "--eval", "--num-workers", "0",
                    "--chkpt-dir", str(tmp / "c"), "--image-dir", str(tmp / "i")])


def test_eval_flag_scores_fid_with_a_provisioned_feature_network(emu, workdir, monkeypatch):
    """--eval as the reference wires it (train.py:204-211, utils/train.py:225-226): the Evaluator samples eval_total_size images through
    the trainer's sample_fn and the FID lands in the checkpoint.  The feature network is a TorchScript file named by
    DDPM_TORCH_AMD_INCEPTION (here a toy stand-in with the 2048-wide output of the real one) and the dataset statistics come from
    precomputed/fid_stats_cifar10_train.npz, the file the reference would download."""
    import numpy as np
    tmp, cfg = workdir

    class Head(torch.nn.Module):
        def forward(self, x):
            return x.mean(dim=(2, 3)).repeat(1, 683)[:, :2048]

    torch.jit.script(Head()).save(str(tmp / "head.pt"))
    monkeypatch.setenv("DDPM_TORCH_AMD_INCEPTION", str(tmp / "head.pt"))
    os.makedirs(tmp / "precomputed")
    np.savez(tmp / "precomputed" / "fid_stats_cifar10_train.npz", mu=np.zeros(2048), sigma=np.eye(2048))
    train = _import_cli("train")
    train.main(["--config-path", cfg, "--train-device", "cpu", "--eval-device", "cpu", "--dry-run", "--num-samples", "0", "--eval", "--use-ddim",
                "--subseq-size", "4", "--eval-total-size", "6", "--eval-batch-size", "4", "--num-workers", "0", "--compute", "fp32",
                "--chkpt-dir", str(tmp / "c"), "--image-dir", str(tmp / "i")])
    saved = torch.load(tmp / "c" / "tiny" / "tiny_1.pt", weights_only=False)
    assert "fid" in saved and saved["fid"] is not None and saved["fid"] > 0 and saved["fid"] == saved["fid"]


@pytest.mark.skipif(not os.path.exists("/root/reference/train.py"), reason="the reference checkout only exists in the build container")
def test_the_reference_train_script_runs_unmodified_against_this_package(emu, workdir, monkeypatch):
    """`from ddpm_torch import *` / `from ddim import *` of the upstream train.py resolve to this package; --dry-run performs one
    update through the engine and writes a checkpoint in the upstream layout."""
    tmp, cfg = workdir
    for m in [k for k in sys.modules if k in ("train", "ddim") or k.startswith("ddpm_torch.")]:
        pass                                                # (modules already imported from this package are the ones upstream gets)
    monkeypatch.setattr(sys, "path", [PKG] + [p for p in sys.path if "reference" not in p])
    monkeypatch.setattr(sys, "argv", ["train.py", "--config-path", cfg, "--train-device", "cpu", "--eval-device", "cpu", "--dry-run",
                                      "--num-samples", "0", "--num-workers", "0", "--chkpt-dir", str(tmp / "refchk"), "--image-dir", str(tmp / "refimg")])
    emu.log.clear()
    runpy.run_path("/root/reference/train.py", run_name="__main__")
    chk = tmp / "refchk" / "tiny" / "tiny_1.pt"
    assert chk.exists()
    saved = torch.load(chk, weights_only=False)
    assert set(saved) >= {"model", "optimizer", "ema", "scheduler", "epoch"} and saved["epoch"] == 1
    assert "ddpm_conv2d_nhwc" in emu.log and "ddpm_mt_adam_ema" in emu.log          # the engine did the work


@pytest.mark.skipif(not os.path.exists("/root/reference/generate.py"), reason="the reference checkout only exists in the build container")
def test_the_reference_generate_script_runs_unmodified_against_this_package(emu, workdir, monkeypatch):
    tmp, cfg = workdir
    train = _import_cli("train")
    train.main(["--config-path", cfg, "--train-device", "cpu", "--dry-run", "--num-samples", "0", "--num-workers", "0", "--compute", "fp32",
                "--chkpt-dir", str(tmp / "chk"), "--image-dir", str(tmp / "img")])
    chk = tmp / "chk" / "tiny" / "tiny_1.pt"
    monkeypatch.setattr(sys, "path", [PKG] + [p for p in sys.path if "reference" not in p])
    monkeypatch.setattr(sys, "argv", ["generate.py", "--config-path", cfg, "--device", "cpu", "--chkpt-path", str(chk), "--use-ddim", "--subseq-size", "4",
                                      "--total-size", "2", "--batch-size", "2", "--save-dir", str(tmp / "refgen")])
    runpy.run_path("/root/reference/generate.py", run_name="__main__")
    assert len(glob.glob(str(tmp / "refgen" / "eval" / "tiny" / "tiny_1" / "*.png"))) == 2
