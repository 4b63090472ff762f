"""-m gpu, needs >= 2 visible GPUs (skipped on the 1-GPU boxes; the driver's 8-GPU node runs it): one rank per GPU over
the "nccl" (= RCCL) backend, real kernels.  Same checks as the 2-rank gloo test on CPU (tests/test_ddp_gloo.py) —
broadcast parameters, averaged gradients equal to the oracle's mean, torch DDP wrapper equivalent — plus the captured
training step whose all-reduces run between hipGraph segments against the eager step."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import unet_ref as U
from tests._ddp_worker import TINY

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _world():
    return min(torch.cuda.device_count(), 8)


def _launch(mode, out_dir, world, **extra_env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(r), str(world), str(port), mode, str(out_dir), "cuda"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    return [torch.load(os.path.join(out_dir, f"{mode}_{r}.pt"), weights_only=True) for r in range(world)]


def _expected(recs):
    sd0, acc = recs[0]["sd"], None
    for r in recs:
        p = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        (U.unet_forward(p, TINY, r["x"], r["t"], training=True) * r["gy"]).sum().backward()
        acc = {k: v.grad for k, v in p.items()} if acc is None else {k: acc[k] + p[k].grad for k in acc}
    return {k: v / len(recs) for k, v in acc.items()}


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
@pytest.mark.parametrize("mode", ["native", "ddp"])
def test_multi_rank_gradients_over_rccl(tmp_path, mode):
    world = _world()
    recs = _launch(mode, tmp_path, world)
    for r in recs[1:]:
        for k in recs[0]["sd"]:
            assert torch.equal(recs[0]["sd"][k], r["sd"][k]), f"{k}: parameters not broadcast"
    want = _expected(recs)
    for k, g0 in recs[0]["grads"].items():
        for r in recs[1:]:
            assert float((g0 - r["grads"][k]).abs().max()) < 1e-6, f"{k}: ranks disagree"
        scale = max(float(want[k].abs().max()), 1e-4)
        assert float((g0 - want[k]).abs().max()) <= 1e-3 * scale + 1e-5, k


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
def test_captured_data_parallel_step_equals_eager(tmp_path):
    """4 distributed Trainer.steps: graph segments with the RCCL all-reduces issued between them == the eager direct step;
    replicas stay in lock-step."""
    world = _world()
    (tmp_path / "g").mkdir(); (tmp_path / "e").mkdir()
    _launch("native", tmp_path / "g", world)
    _launch("native_eager", tmp_path / "e", world)
    g = [torch.load(os.path.join(tmp_path / "g", f"after_step_native_{r}.pt"), weights_only=True) for r in range(world)]
    e = [torch.load(os.path.join(tmp_path / "e", f"after_step_native_eager_{r}.pt"), weights_only=True) for r in range(world)]
    assert g[0]["direct"] and g[0]["segments"] is not None and g[0]["segments"] >= 2, g[0]["segments"]    # >= 1 cut for the tail all-reduce
    assert e[0]["segments"] is None
    for r in range(1, world):
        for k in g[0]["sd"]:
            assert float((g[0]["sd"][k] - g[r]["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged"
    for k in g[0]["sd"]:
        scale = float(e[0]["sd"][k].abs().max()) or 1.0
        assert float((g[0]["sd"][k] - e[0]["sd"][k]).abs().max()) <= 2e-3 * scale + 2e-4, k
    assert g[0]["losses"] == pytest.approx(e[0]["losses"], rel=1e-3)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
def test_cifar_geometry_native_exchange_equals_torch_ddp(tmp_path):
    """BASELINE config 3's geometry (configs/cifar10.json at 32 x 32, B = 4 per rank; upstream train.py:110, utils/train.py:166-169):
    the native chunked all-reduce (6 chunks + tail of the packed staging buffer, issued inside the backward) gives every rank the
    gradients torch's DistributedDataParallel gives, the ranks agree bit for bit, and the distributed Trainer.steps on top keep the
    replicas in lock-step."""
    world = _world()
    (tmp_path / "n").mkdir(); (tmp_path / "d").mkdir()
    nat = _launch("native", tmp_path / "n", world, DDP_WORKER_CFG="cifar", DDPM_TORCH_AMD_COMPUTE="fp32")
    ddp = _launch("ddp", tmp_path / "d", world, DDP_WORKER_CFG="cifar", DDPM_TORCH_AMD_COMPUTE="fp32")
    for k, g0 in nat[0]["grads"].items():
        for r in nat[1:]:
            assert torch.equal(g0, r["grads"][k]), f"{k}: ranks disagree after the all-reduce"
        scale = max(float(ddp[0]["grads"][k].abs().max()), 1e-6)
        assert float((g0 - ddp[0]["grads"][k]).abs().max()) <= 1e-4 * scale + 1e-7, k
    after = [torch.load(os.path.join(tmp_path / "n", f"after_step_native_{r}.pt"), weights_only=True) for r in range(world)]
    for r in range(1, world):
        for k in after[0]["sd"]:
            assert float((after[0]["sd"][k] - after[r]["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
def test_celebahq_geometry_native_exchange_equals_torch_ddp(tmp_path):
    """BASELINE config 5's geometry (configs/celebahq.json at 256 x 256, B = 1 per rank here): 454.7 MB of packed gradients per exchange,
    cut into chunks and all-reduced from inside the backward — every rank ends with the gradients torch's DistributedDataParallel gives
    (upstream train.py:110), the ranks agree bit for bit, and distributed Trainer.steps keep the replicas in lock-step."""
    world = _world()
    (tmp_path / "n").mkdir(); (tmp_path / "d").mkdir()
    nat = _launch("native", tmp_path / "n", world, DDP_WORKER_CFG="celebahq", DDPM_TORCH_AMD_COMPUTE="fp32")
    ddp = _launch("ddp", tmp_path / "d", world, DDP_WORKER_CFG="celebahq", DDPM_TORCH_AMD_COMPUTE="fp32")
    for k, g0 in nat[0]["grads"].items():
        for r in nat[1:]:
            assert torch.equal(g0, r["grads"][k]), f"{k}: ranks disagree after the all-reduce"
        scale = max(float(ddp[0]["grads"][k].abs().max()), 1e-6)
        assert float((g0 - ddp[0]["grads"][k]).abs().max()) <= 1e-4 * scale + 1e-7, k
    after = [torch.load(os.path.join(tmp_path / "n", f"after_step_native_{r}.pt"), weights_only=True) for r in range(world)]
    for r in range(1, world):
        for k in after[0]["sd"]:
            assert float((after[0]["sd"][k] - after[r]["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
@pytest.mark.parametrize("reserved", ["0", "16"])
def test_launch_plan_data_parallel_step_equals_eager(tmp_path, reserved):
    """The launch-plan form of the distributed step (csrc/plan.hip; the RCCL all-reduces are host callbacks between plan segments), also
    with compute units held back for the communicator (DDPM_DP_RESERVED_CUS): same parameters as the eager step, replicas in lock-step."""
    world = _world()
    (tmp_path / "p").mkdir(); (tmp_path / "e").mkdir()
    _launch("native_plan", tmp_path / "p", world, DDPM_DP_RESERVED_CUS=reserved)
    _launch("native_eager4", tmp_path / "e", world, DDPM_DP_RESERVED_CUS=reserved)
    p = [torch.load(os.path.join(tmp_path / "p", f"after_step_native_plan_{r}.pt"), weights_only=True) for r in range(world)]
    e = [torch.load(os.path.join(tmp_path / "e", f"after_step_native_eager4_{r}.pt"), weights_only=True) for r in range(world)]
    assert p[0]["direct"] and p[0]["last_kind"] == "plan" and p[0]["plan_segments"] >= 2
    for r in range(1, world):
        for k in p[0]["sd"]:
            assert float((p[0]["sd"][k] - p[r]["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged"
    for k in p[0]["sd"]:
        scale = float(e[0]["sd"][k].abs().max()) or 1.0
        assert float((p[0]["sd"][k] - e[0]["sd"][k]).abs().max()) <= 2e-3 * scale + 2e-4, k
    assert p[0]["losses"] == pytest.approx(e[0]["losses"], rel=1e-3)
