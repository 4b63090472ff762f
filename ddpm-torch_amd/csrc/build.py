"""Build csrc/libddpm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python ddpm-torch_amd/csrc/build.py [--force]

One shared object, plain C ABI (include/ddpm_hip.h), no torch / pybind types.  The .so is git-ignored but travels
with the tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "wgrad.hip", "wgrad1x1.hip", "attention.hip", "pointwise.hip", "conv3x3.hip", "edgeconv.hip", "norm.hip", "elementwise.hip", "optim.hip", "probe.hip", "plan.hip"]
OUT = os.path.join(HERE, "libddpm_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr = os.path.join(HERE, "common.h")
    inc = os.path.join(HERE, "..", "..", "include")
    abi = [os.path.join(inc, "ddpm_hip.h"), os.path.join(inc, "ddpm_hip_debug.h")]      # plan.hip generates its call thunks from these declarations
    objs = []
    procs = []
    for s in SOURCES:
        src, obj = os.path.join(HERE, s), os.path.join(HERE, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, hdr, *abi]):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
