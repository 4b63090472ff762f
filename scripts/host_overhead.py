"""How much of a step is host (Python + ctypes launch) time?  Times eval forward and a full train step at B=128 and B=4:
at B=4 the GPU work is tiny, so the wall time ~= host time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import ddpm_torch
from bench import CIFAR
dev = "cuda:0"
torch.manual_seed(0)
m = ddpm_torch.UNet(**CIFAR).to(dev).set_compute_dtype("bf16")
dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
opt = torch.optim.Adam(m.parameters(), lr=2e-4)
tr = ddpm_torch.Trainer(m, opt, dif, epochs=1, trainloader=None, use_ema=True, shape=(3, 32, 32), device=torch.device(dev))
for B in (128, 4):
    x = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
    t = torch.randint(0, 1000, (B,), device=dev)
    m.eval()
    with torch.inference_mode():
        for _ in range(5): m(x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): m(x, t)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"B={B:4d} eval forward: host issue {t_issue / 30 * 1e3:6.2f} ms, wall {t_all / 30 * 1e3:6.2f} ms per call")
    m.train()
    for _ in range(3): tr.step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.step(x)
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"B={B:4d} train step  : wall {t_all / 10 * 1e3:6.2f} ms per step")

if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    x = torch.rand(4, 3, 32, 32, device=dev) * 2 - 1
    m.train()
    for _ in range(3): tr.step(x)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10): tr.step(x)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
