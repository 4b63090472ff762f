"""Sampling throughput vs batch (CIFAR-10 UNet, bf16, graph-replayed step): how far does a 288-GB part let the batch grow, and does it pay?
Prints ms per step, samples/s for 1000-step chains and peak memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch, ddpm_torch
from bench import CIFAR
S = 24
torch.manual_seed(0)
m = ddpm_torch.UNet(**CIFAR).to("cuda:0").set_compute_dtype("bf16").eval()
dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, S), "eps", "fixed-large", "mse")
for B in [int(v) for v in (sys.argv[1:] or ["128", "512", "1024", "2048", "4096"])]:
    torch.cuda.reset_peak_memory_stats()
    try:
        dif.p_sample(m, shape=(B, 3, 32, 32), device="cuda:0", seed=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = dif.p_sample(m, shape=(B, 3, 32, 32), device="cuda:0", seed=2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / S
        assert torch.isfinite(x).all()
        print(f"B={B:5d}: {dt * 1e3:8.3f} ms/step  {B / (dt * 1000):7.2f} samples/s (1000-step chains)  {B * 12.444e9 / dt / 1e12:6.1f} TFLOP/s  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    except Exception as e:
        print(f"B={B}: {type(e).__name__}: {e}", flush=True)
        break
