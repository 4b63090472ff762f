import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch, ddpm_torch
from bench import CIFAR
S = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
m = ddpm_torch.UNet(**CIFAR).to("cuda:0").set_compute_dtype(os.environ.get("DT", "bf16")).eval()
dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, S), "eps", "fixed-large", "mse")
dif.p_sample(m, shape=(int(os.environ.get("WARM_B", "8")), 3, 32, 32), device="cuda:0", seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
x = dif.p_sample(m, shape=(128, 3, 32, 32), device="cuda:0", seed=2)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / S * 1e3)
