"""Debug aid: one-rank RCCL communicator, a few eager steps, then force the hipGraph form and print the full traceback of a failed capture."""
import os, socket, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
import ddpm_torch
import ddpm_torch.utils.train as train_mod
from ddpm_torch import _graphs
from bench import CIFAR, make_trainer

orig = _graphs.SegmentedGraph._capture
def verbose(self, body):
    try:
        return orig(self, body)
    except BaseException:
        traceback.print_exc()
        raise
_graphs.SegmentedGraph._capture = verbose
train_mod._TRAIN_GRAPH = os.environ.get("FIRST_FORM", "0") if os.environ.get("FIRST_FORM", "0") == "plan" else False
ddpm_torch.seed_all(1234)
model, net, dif, tr = make_trainer(ddpm_torch, CIFAR, dev, "bf16", (3, 32, 32), "fixed-large", True, 0, True, 0)
x0 = (torch.rand(16, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
net.train()
for i in range(4):
    tr.step(x0, global_steps=i + 1)
torch.cuda.synchronize()
print("eager/plan steps done", flush=True)
train_mod._TRAIN_GRAPH = True
for i in range(3):
    tr.step(x0, global_steps=10 + i)
torch.cuda.synchronize()
ds = next(iter(tr._direct.values()))
print("graph steps done; last_kind", ds.last_kind, "graph_failed", ds.graph_failed, flush=True)
