"""-m gpu, ONE GPU: the data-parallel path with world_size 2 on real kernels.  Both ranks share GPU 0 and exchange over gloo (RCCL refuses
two ranks on one device), so this is not a statement about xGMI — it runs what tests/test_ddp_gloo.py runs on the CPU emulator (broadcast
parameters, chunked all-reduce issued from inside the hand-written backward, mean-gradient semantics, torch DDP equivalence, distributed
Trainer.steps) through the HIP library, the two streams and the launch plan (the all-reduces between plan segments) instead.  The N-GPU
RCCL form of the same checks is tests/test_multi_gpu.py (needs >= 2 GPUs)."""
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
WORLD = 2


def _launch(mode, out_dir, **extra_env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(r), str(WORLD), str(port), mode, str(out_dir), "cuda_shared"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(WORLD)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    return [torch.load(os.path.join(out_dir, f"{mode}_{r}.pt"), weights_only=True) for r in range(WORLD)]


def _expected(recs):
    sd0, acc = recs[0]["sd"], None
    for r in recs:
        p = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        (U.unet_forward(p, TINY, r["x"], r["t"], training=True) * r["gy"]).sum().backward()
        acc = {k: v.grad for k, v in p.items()} if acc is None else {k: acc[k] + p[k].grad for k in acc}
    return {k: v / len(recs) for k, v in acc.items()}


@pytest.mark.parametrize("mode", ["native", "ddp"])
def test_two_ranks_on_one_gpu_average_their_gradients(tmp_path, mode):
    recs = _launch(mode, tmp_path)
    for k in recs[0]["sd"]:
        assert torch.equal(recs[0]["sd"][k], recs[1]["sd"][k]), f"{k}: parameters not broadcast"
    want = _expected(recs)
    for k, g0 in recs[0]["grads"].items():
        assert float((g0 - recs[1]["grads"][k]).abs().max()) < 1e-6, f"{k}: ranks disagree"
        scale = max(float(want[k].abs().max()), 1e-4)
        assert float((g0 - want[k]).abs().max()) <= 1e-3 * scale + 1e-5, k
    if mode == "native":
        a, b = (torch.load(os.path.join(tmp_path, f"after_step_native_{r}.pt"), weights_only=True) for r in range(WORLD))
        assert a["direct"] and a["segments"] is not None and a["segments"] >= 2      # the captured form: all-reduces between graph segments
        for k in a["sd"]:
            assert float((a["sd"][k] - b["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged after distributed Trainer.steps"


def test_two_ranks_launch_plan_steps_equal_eager_steps(tmp_path):
    (tmp_path / "p").mkdir(); (tmp_path / "e").mkdir()
    _launch("native_plan", tmp_path / "p")
    _launch("native_eager4", tmp_path / "e")
    p = [torch.load(os.path.join(tmp_path / "p", f"after_step_native_plan_{r}.pt"), weights_only=True) for r in range(WORLD)]
    e = [torch.load(os.path.join(tmp_path / "e", f"after_step_native_eager4_{r}.pt"), weights_only=True) for r in range(WORLD)]
    assert p[0]["direct"] and p[0]["last_kind"] == "plan" and p[0]["plan_segments"] >= 2 and e[0]["last_kind"] == "eager"
    for k in p[0]["sd"]:
        assert float((p[0]["sd"][k] - p[1]["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged"
        scale = float(e[0]["sd"][k].abs().max()) or 1.0
        assert float((p[0]["sd"][k] - e[0]["sd"][k]).abs().max()) <= 2e-3 * scale + 2e-4, k
    assert p[0]["losses"] == pytest.approx(e[0]["losses"], rel=1e-3)


def test_two_ranks_cifar_geometry_native_exchange_equals_torch_ddp(tmp_path):
    """BASELINE config 3's geometry (configs/cifar10.json at 32 x 32, B = 4 per rank, fp32 mode): the ramped chunk plan (8 chunks + tail, the
    time-projection rows inside the chunks) against torch's DistributedDataParallel, on device."""
    (tmp_path / "n").mkdir(); (tmp_path / "d").mkdir()
    nat = _launch("native", tmp_path / "n", DDP_WORKER_CFG="cifar", DDPM_TORCH_AMD_COMPUTE="fp32")
    ddp = _launch("ddp", tmp_path / "d", DDP_WORKER_CFG="cifar", DDPM_TORCH_AMD_COMPUTE="fp32")
    for k, g0 in nat[0]["grads"].items():
        assert torch.equal(g0, nat[1]["grads"][k]), f"{k}: ranks disagree after the all-reduce"
        scale = max(float(ddp[0]["grads"][k].abs().max()), 1e-6)
        assert float((g0 - ddp[0]["grads"][k]).abs().max()) <= 1e-4 * scale + 1e-7, k
    after = [torch.load(os.path.join(tmp_path / "n", f"after_step_native_{r}.pt"), weights_only=True) for r in range(WORLD)]
    for k in after[0]["sd"]:
        assert float((after[0]["sd"][k] - after[1]["sd"][k]).abs().max()) < 1e-7, f"{k}: replicas diverged"
