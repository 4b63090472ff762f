"""-m gpu: the command-line counterparts (reference train.py:141-144, generate.py:72-93,128) end to end ON THE
DEVICE through the real C ABI: `train.py --dry-run` -> checkpoint -> resume -> `generate.py` -> PNG files, and the forward of the
re-loaded EMA weights against the oracle."""
import glob
import importlib
import json
import os
import sys

import pytest
import torch

from oracle import unet_ref as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ddpm-torch_amd")
DEV = "cuda:0"

CFG = {
    "dataset": "cifar10",
    "diffusion": {"timesteps": 40, "beta_start": 1e-4, "beta_end": 0.02, "beta_schedule": "linear", "model_mean_type": "eps",
                  "model_var_type": "fixed-large", "loss_type": "mse"},
    "model": {"in_channels": 3, "hid_channels": 32, "ch_multipliers": [1, 2], "num_res_blocks": 1, "apply_attn": [False, True], "drop_rate": 0.1},
    "train": {"lr": 2e-4, "batch_size": 4, "grad_norm": 1.0, "epochs": 1, "warmup": 10, "use_ema": True, "ema_decay": 0.9999},
}


def _cli(name):
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    return importlib.import_module(name)


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_train_dry_run_checkpoint_generate_on_device(tmp_path, monkeypatch, compute):
    cfg = tmp_path / "tiny.json"
    cfg.write_text(json.dumps(CFG))
    monkeypatch.setenv("DDPM_TORCH_AMD_SYNTHETIC_DATA", "8")
    monkeypatch.chdir(tmp_path)
    train, gen = _cli("train"), _cli("generate")
    tr = train.main(["--config-path", str(cfg), "--train-device", DEV, "--dry-run", "--num-samples", "4", "--num-workers", "0",
                     "--compute", compute, "--chkpt-dir", str(tmp_path / "chk"), "--image-dir", str(tmp_path / "img")])
    assert tr.ema.num_updates == 0 and int(next(iter(tr.optimizer.state.values()))["step"]) == 1          # exactly one update
    assert next(tr.model.parameters()).is_cuda and tr._direct, "the dry run must have gone through the direct (C-ABI) step"
    chk = tmp_path / "chk" / "tiny" / "tiny_1.pt"
    assert chk.exists() and (tmp_path / "img" / "train" / "tiny" / "1.jpg").exists()
    saved = torch.load(chk, map_location="cpu", weights_only=False)
    assert set(saved) >= {"model", "optimizer", "ema", "scheduler", "epoch"} and saved["epoch"] == 1
    assert list(saved["model"]) == list(U.init_state_dict(dict(CFG["model"], out_channels=3)))             # reference key order (Appendix B)
    moved = max(float((saved["model"][k] - saved["ema"]["shadow"][k]).abs().max()) for k in saved["ema"]["shadow"])
    assert 0 < moved < 1e-2                                      # one Adam step at lr 2e-5 (warm-up) away from the shadow's initial copy
    # resume on the device: epoch counter, EMA counter and Adam state come back, and the next step runs
    tr2 = train.main(["--config-path", str(cfg), "--train-device", DEV, "--dry-run", "--num-samples", "0", "--resume", "--num-workers", "0",
                      "--compute", compute, "--chkpt-path", str(chk), "--chkpt-dir", str(tmp_path / "chk2"), "--image-dir", str(tmp_path / "img2")])
    assert tr2.ema.num_updates == 1 and int(next(iter(tr2.optimizer.state.values()))["step"]) == 2
    # generate.py: EMA shadow preferred, frozen model, PNG files; --seed makes whole chains repeat
    outs = []
    for rep in range(2):
        out = gen.main(["--config-path", str(cfg), "--device", DEV, "--chkpt-path", str(chk), "--total-size", "5", "--batch-size", "4",
                        "--save-dir", str(tmp_path / f"gen{rep}"), "--compute", compute, "--seed", "7"])
        files = sorted(glob.glob(os.path.join(out, "*.png")))
        assert len(files) == 5
        from PIL import Image
        import numpy as np
        imgs = np.stack([np.asarray(Image.open(f)) for f in files])
        assert imgs.shape == (5, 32, 32, 3) and imgs.std() > 0
        outs.append(np.sort(imgs.reshape(5, -1).astype(np.int64).sum(1)))        # file names are random uuids: compare as a multiset
    assert (outs[0] == outs[1]).all()
    # the re-loaded EMA weights, forward on the device vs the oracle on the host
    process, model, device, shape, _ = gen.build(gen.parse_args(["--config-path", str(cfg), "--device", DEV, "--chkpt-path", str(chk), "--compute", compute]))
    assert not any(p.requires_grad for p in model.parameters()) and not model.training
    sd = {k: v.cpu() for k, v in saved["ema"]["shadow"].items()}
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(2, 3, 32, 32, generator=g), torch.tensor([1, 38])
    with torch.no_grad():
        y = model(x.to(DEV), t.to(DEV)).cpu()
        ref = U.unet_forward(sd, dict(CFG["model"], out_channels=3), x, t, training=False)
    rel = float((y - ref).abs().max() / ref.abs().max())
    assert rel < (1e-3 if compute == "fp32" else 6e-2), rel
