"""CPU, world_size 2, gloo: the N>1 training path.  Each rank runs the unmodified engine (C-ABI emulated) on its own shard.
Checks: (1) native path — parameters broadcast from rank 0, gradients after the in-backward chunked all-reduce are identical
on both ranks and equal the MEAN of the per-rank gradients (DDP semantics, train.py:110); (2) the reference's own wrapper
``DistributedDataParallel(model)`` gives the same gradients; (3) a distributed Trainer.step keeps the replicas in lock-step."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from ddpm_torch import _hip
from oracle import unet_ref as U
from tests._ddp_worker import TINY

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, out_dir):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(r), "2", str(port), mode, str(out_dir)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return [torch.load(os.path.join(out_dir, f"{mode}_{r}.pt"), weights_only=True) for r in range(2)]


def _expected(recs):
    """mean over ranks of the oracle's gradients, all ranks starting from rank 0's weights"""
    sd0 = recs[0]["sd"]
    acc = None
    for r in recs:
        p = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        y = U.unet_forward(p, TINY, r["x"], r["t"], training=True)
        (y * r["gy"]).sum().backward()
        acc = {k: v.grad for k, v in p.items()} if acc is None else {k: acc[k] + p[k].grad for k in acc}
    return {k: v / len(recs) for k, v in acc.items()}


@pytest.mark.parametrize("mode", ["native", "ddp"])
def test_two_rank_gradients_are_averaged(tmp_path, mode):
    recs = _launch(mode, tmp_path)
    for k in recs[0]["sd"]:
        assert torch.equal(recs[0]["sd"][k], recs[1]["sd"][k]), f"{k}: parameters not broadcast"
    want = _expected(recs)
    for k, g0 in recs[0]["grads"].items():
        assert torch.allclose(g0, recs[1]["grads"][k], rtol=0, atol=0) or float((g0 - recs[1]["grads"][k]).abs().max()) < 1e-7, k
        scale = max(float(want[k].abs().max()), 1e-4)
        assert float((g0 - want[k]).abs().max()) <= 3e-4 * scale + 5e-6, k      # analytically-zero grads (bias under GroupNorm) are fp32 noise
    if mode == "native":
        a, b = (torch.load(os.path.join(tmp_path, f"after_step_native_{r}.pt"), weights_only=True) for r in range(2))
        assert a["direct"] and b["direct"]                    # the autograd-free step ran (its backward issues the same all-reduces)
        for k in a["sd"]:
            assert float((a["sd"][k] - b["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged after a distributed Trainer.step"


def test_two_rank_launch_plan_steps_equal_eager_steps(tmp_path):
    """The recorded form of the distributed step (csrc/plan.hip's table, here replayed through the emulator): the RCCL / gloo all-reduces sit
    BETWEEN plan segments as host callbacks.  Four steps (eager warm-up, recording, two replays) must leave both ranks exactly where four
    eager steps leave them."""
    _launch("native_plan", tmp_path)
    _launch("native_eager4", tmp_path)
    for r in range(2):
        p = torch.load(os.path.join(tmp_path, f"after_step_native_plan_{r}.pt"), weights_only=True)
        e = torch.load(os.path.join(tmp_path, f"after_step_native_eager4_{r}.pt"), weights_only=True)
        assert p["direct"] and p["last_kind"] == "plan" and e["last_kind"] == "eager"
        assert p["plan_segments"] >= 2 and p["plan_launches"] > 100           # at least one exchange inside the backward + the tail
        assert p["losses"] == e["losses"]
        for k in e["sd"]:
            assert torch.equal(p["sd"][k], e["sd"][k]), k
        for k in e["shadow"]:
            assert torch.equal(p["shadow"][k], e["shadow"][k]), k
