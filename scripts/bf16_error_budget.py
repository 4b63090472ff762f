"""Where does the bf16 mode's forward error come from?  (CPU; no GPU needed.)

The engine's throughput mode stores activations and packed conv weights in bf16 and accumulates in fp32; its forward differs from the
reference's fp32 forward by 1.5e-2 of the output range on BASELINE config 2 at B = 128 (tests/test_config2_bench_batch_gpu.py, fixture G11).
This script re-runs the CPU restatement (oracle/unet_ref.py) with bf16 rounding injected at the places the engine rounds — one class at a
time — and measures each variant against the SAME reference-written fixture, so that the 1.5e-2 has owners:

    weights      every conv weight rounded to bf16 (Linear / GroupNorm parameters and biases stay fp32, as in the engine)
    input        the image rounded to bf16 (the engine's channel-padded NHWC copy)
    conv         conv outputs rounded when stored (after the fused bias / time bias / residual epilogue)
    gn           GroupNorm(+SiLU) outputs rounded when stored
    attn         the attention core's probability tile and output rounded

Output: profiles/r05_bf16_error_budget.txt.  Samples: the first N of the fixture's 128 (default 32; BUDGET_SAMPLES=128 for all).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from oracle import unet_ref as U
from tests.golden.recipes import rnd

g = torch.load(os.path.join(ROOT, "tests", "golden", "g11_config2_b128.pt"), weights_only=True)
N = int(os.environ.get("BUDGET_SAMPLES", "32"))
torch.set_num_threads(int(os.environ.get("BUDGET_THREADS", "8")))
cfg = g["cfg"]
torch.manual_seed(g["init_seed"])
sd = U.randomize_state_dict(U.init_state_dict(cfg), g["rand_seed"])
x = rnd(g["B"], 3, 32, 32, seed=g["fwd"]["x_seed"])[:N]
t = g["fwd"]["t"][:N]
want = g["fwd"]["y_sub"][:N]
scale = g["fwd"]["y_absmax"]
bf = lambda v: v.to(torch.bfloat16).to(torch.float32)
sd_bf = {k: (bf(v) if v.ndim == 4 else v) for k, v in sd.items()}           # conv weights only


def run(name, weights, kinds):
    U.ROUND = {k: bf for k in kinds} or None
    t0 = time.perf_counter()
    with torch.no_grad():
        y = U.unet_forward(sd_bf if weights else sd, cfg, x, t)
    U.ROUND = None
    d = (y[:, :, ::4, ::4] - want).abs()
    print(f"{name:58s} max {float(d.max()) / scale:.3e}   mean {float(d.mean()) / scale:.3e}   ({time.perf_counter() - t0:.0f} s)", flush=True)
    return float(d.max()) / scale


print(f"bf16 error budget of the forward, configs/cifar10.json at 32 x 32, first {N} samples of fixture G11 (errors relative to the output's largest magnitude {scale:.3f})")
print("measured on the GPU, bf16 mode, all 128 samples: max 1.5e-2 (tests/test_config2_bench_batch_gpu.py)")
run("fp32 everywhere (the oracle itself)", False, ())
run("bf16 conv weights only", True, ())
run("bf16 input image only", False, ("input",))
run("bf16 conv outputs only", False, ("conv",))
run("bf16 GroupNorm(+SiLU) outputs only", False, ("gn",))
run("bf16 attention tile + output only", False, ("attn",))
run("all activations bf16, fp32 weights", False, ("input", "conv", "gn", "attn"))
run("everything the engine rounds (weights + all activations)", True, ("input", "conv", "gn", "attn"))
run("... with GroupNorm outputs kept fp32", True, ("input", "conv", "attn"))
run("... with conv outputs kept fp32", True, ("input", "gn", "attn"))
