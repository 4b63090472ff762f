"""BASELINE config 5, per-GPU work: CelebA-HQ 256x256 UNet (configs/celebahq.json), B = 2, full Trainer.step — for kernel traces.
    python scripts/hq_step.py [steps] [train|sample]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
import bench
import ddpm_torch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
dev = torch.device("cuda", 0)
ddpm_torch.seed_all(1234)
m, net, dif, tr = bench.make_trainer(ddpm_torch, bench.CELEBAHQ, dev, "bf16", (3, 256, 256), "fixed-small", lr=2e-5)
if mode == "train":
    net.train()
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)
    el = bench.timed_steps(tr, x, steps, 4, torch.cuda.synchronize)
    print(f"celebahq B=2 train: {el / steps * 1e3:.2f} ms/step ({2 * steps / el:.1f} img/s)")
else:
    net.eval()
    d = ddpm_torch.GaussianDiffusion(ddpm_torch.get_beta_schedule("linear", 1e-4, 0.02, steps), "eps", "fixed-small", "mse")
    d.prepare_sampler(m, (8, 3, 256, 256), dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.p_sample(m, shape=(8, 3, 256, 256), device=dev, seed=1)
    torch.cuda.synchronize()
    print(f"celebahq B=8 sampling: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
