"""Host time per training step, by form of the step (eager launches | launch plan | hipGraph replay), CIFAR-10 UNet, bf16, B = 128.

Method: the GPU is parked behind a spin kernel while the host enqueues N whole steps, so the time until ``Trainer.step`` returns is pure
host work (interpreter + ctypes + HIP runtime enqueue), not GPU back-pressure; the loss read-back of the previous step is dropped before
each call (it would wait for the parked GPU).  Then the same N steps are timed end to end with the GPU running (wall per step, pipelined).
Output is committed as profiles/r05_host_overhead.txt.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import ddpm_torch
from ddpm_torch.utils import train as train_mod
from bench import CIFAR

dev = "cuda:0"
N = int(os.environ.get("HOST_STEPS", "6"))
B = int(os.environ.get("HOST_BATCH", "128"))
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
cycles_per_ms = 20_000_000 / ((time.perf_counter() - t0) * 1e3)

for form in (False, "plan", True):
    train_mod._TRAIN_GRAPH = form
    torch.manual_seed(0)
    m = ddpm_torch.UNet(**CIFAR).to(dev).set_compute_dtype("bf16")
    dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, grad_norm=1.0, shape=(3, 32, 32), device=torch.device(dev))
    x = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
    m.train()
    for i in range(6):
        tr.step(x, global_steps=i + 1)
    torch.cuda.synchronize()
    ds = next(iter(tr._direct.values()))
    # (1) host only: GPU parked for ~N * 12 ms + slack
    tr._collect_loss()
    torch.cuda._sleep(int(cycles_per_ms * (N * 14 + 40)))
    t0 = time.perf_counter()
    for i in range(N):
        tr._loss_pending = None
        tr.step(x, global_steps=7 + i)
    host = (time.perf_counter() - t0) / N
    tr._loss_pending = None
    torch.cuda.synchronize()
    # (2) wall, pipelined
    for i in range(3):
        tr.step(x, global_steps=20 + i)
    tr.current_stats; torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        tr.step(x, global_steps=30 + i)
    tr.current_stats; torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    name = {False: "eager launches", "plan": "launch plan", True: "hipGraph replay"}[form]
    extra = f", {ds.plan.launches} recorded calls in {len(ds.plan.segments)} segment(s)" if ds.plan is not None else ""
    print(f"B={B} {name:16s}: host {host * 1e3:6.2f} ms per step, wall {wall * 1e3:6.2f} ms per step (form run: {ds.last_kind}{extra})", flush=True)
    del tr, opt, m
