"""Is the sampling step faster re-issued as a launch plan (csrc/plan.hip) than replayed as a hipGraph?  CIFAR-10 UNet, bf16, B = 128 / 512.
One-off measurement (round 5): the training step's graph form loses ~1 ms per step to the executor serialising two stream branches; the
sampling step is a single-stream chain, so the graph has no such handicap — this checks whether it has another."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import ddpm_torch
from ddpm_torch import _hip
from ddpm_torch._plan import LaunchPlan
from ddpm_torch.diffusion import _STEP_TABLES, _MEAN_CODE
from bench import CIFAR

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = ddpm_torch.UNet(**CIFAR).to(dev).set_compute_dtype("bf16").eval()
dif = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, 1000), "eps", "fixed-large", "mse")
STEPS = 300
for B in (128, 512):
    shape = (B, 3, 32, 32)
    with torch.inference_mode():
        for _ in range(2):
            t0 = time.perf_counter(); x = dif.p_sample(m, shape=shape, device=dev, seed=1); torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        # plan form of the same step
        x_t = torch.randn(shape, device=dev)
        z = torch.empty_like(x_t)
        t = torch.full((B,), 999, dtype=torch.int64, device=dev)
        tabs = [dif._tab(n_, dev) for n_ in _STEP_TABLES]
        n = x_t[0].numel()

        def body(cut=None):
            out = m(x_t, t).contiguous().float()
            _hip.call("ddpm_p_sample_step", x_t.data_ptr(), out.data_ptr(), z.data_ptr(), t.data_ptr(), *[tb.data_ptr() for tb in tabs],
                      x_t.data_ptr(), 0, B, n, _MEAN_CODE["eps"], 1, 1000, _hip.stream())
            _hip.call("ddpm_add_i64", t.data_ptr(), B, -1, _hip.stream())
        with dif._time_tables(m):
            body()
            plan = LaunchPlan(dev).record(body)
            assert plan.build_error is None
            t.fill_(999)
            for _ in range(20):
                z.normal_(); plan.replay()
            torch.cuda.synchronize(); t.fill_(999); t0 = time.perf_counter()
            for _ in range(STEPS):
                z.normal_(); plan.replay()
            torch.cuda.synchronize()
            tp = (time.perf_counter() - t0) / STEPS
    print(f"B={B}: hipGraph replay {tg:.3f} s per 1000-step chain = {tg:.3f} ms/step | launch plan {tp * 1e3:.3f} ms/step ({plan.launches} calls)", flush=True)
