"""Per-step wall times of the training step (sync after every step) to see jitter / drift."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch, ddpm_torch
from bench import CIFAR
dev = "cuda:0"
torch.manual_seed(0)
m = ddpm_torch.UNet(**CIFAR).to(dev).set_compute_dtype("bf16")
dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
opt = torch.optim.Adam(m.parameters(), lr=2e-4)
tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 32, 32), device=torch.device(dev))
x = torch.rand(128, 3, 32, 32, device=dev) * 2 - 1
m.train()
for _ in range(5): tr.step(x)
torch.cuda.synchronize()
import gc
if os.environ.get("NOGC"): gc.disable()
if os.environ.get("GCFREEZE"): gc.collect(); gc.freeze()
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    t0 = time.perf_counter(); tr.step(x); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts_s = sorted(ts)
print("per-step ms: min %.2f  p50 %.2f  p90 %.2f  max %.2f" % (ts_s[0], ts_s[len(ts) // 2], ts_s[int(len(ts) * .9)], ts_s[-1]))
print(" ".join("%.1f" % t for t in ts))
print("gc counts", gc.get_count(), "mem", torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats().get("num_sync_all_streams", None))
