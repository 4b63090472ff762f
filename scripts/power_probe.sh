#!/bin/bash
# Sustained launches of one conv shape while rocm-smi samples power and shader clock: is the kernel running into the power cap?
#   usage: power_probe.sh H C N [prev|new]
cd "$GRAFT_REPO_ROOT" || exit 1
python - "$@" <<'PY' &
import ctypes, math, os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "ddpm-torch_amd")]
import torch
from ddpm_torch import _hip
H, C, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
name = "libddpm_hip_prev.so" if len(sys.argv) > 4 and sys.argv[4] == "prev" else "libddpm_hip.so"
h = ctypes.CDLL(os.path.join(ROOT, "ddpm-torch_amd", "csrc", name))
h.ddpm_conv2d_nhwc.argtypes = _hip.PROTOTYPES["ddpm_conv2d_nhwc"]
B, dt, DEV = 128, torch.bfloat16, "cuda:0"
x = torch.randn(B, H, H, C, device=DEV).to(dt)
if os.environ.get("ZERO"): x.zero_()
w = (torch.randn(N, 9 * C, device=DEV) / math.sqrt(9 * C)).to(dt)
y = torch.empty(B, H, H, N, device=DEV, dtype=dt)
bias = torch.zeros(N, device=DEV)
st = torch.cuda.current_stream().cuda_stream
def fn():
    h.ddpm_conv2d_nhwc(x.data_ptr(), C, w.data_ptr(), y.data_ptr(), N, bias.data_ptr(), 0, 0, 0, 0, B, H, H, C, H, H, N, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 1, st)
for _ in range(10): fn()
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10000): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10
    print(f"  [{name}] t={time.time() - t0:4.1f}s {us:6.1f} us/launch {2.0 * B * H * H * N * 9 * C / us / 1e6:6.0f} TF", flush=True)
PY
sleep 1.5
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done
wait
