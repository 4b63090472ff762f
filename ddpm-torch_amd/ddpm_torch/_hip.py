"""ctypes binding of the C-ABI library ``csrc/libddpm_hip.so`` (declared in ``include/ddpm_hip.h``).

The product path has NO fallback: if the shared object is missing or a symbol is absent this module
raises, and every op raises on non-CUDA tensors.  Build with ``python ddpm-torch_amd/csrc/build.py``
(``hipcc --offload-arch=gfx950``); ``__graft_entry__.build()`` does the same.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_ulonglong, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDPM_HIP_LIB: load another build of the library instead (A/B and ablation drivers point it at csrc/libddpm_hip_<name>.so — they must
# never overwrite the product library; scripts/build_variant.sh builds such variants)
LIB_PATH = os.environ.get("DDPM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libddpm_hip.so")

F32, BF16 = 0, 1
_ERR = {1: "bad shape / divisibility", 2: "unsupported dtype", 3: "misaligned pointer or pitch",
        4: "kernel launch failed", 5: "null pointer"}

P, I, L, F, U = c_void_p, c_int, c_longlong, c_float, c_ulonglong

# name -> argtypes; every function returns int status (0 = OK) except ddpm_gn_workspace_floats and
# the ddpm_*_variant queries / ddpm_wgrad_effective_splits / ddpm_conv3x3_wgrad_splits (plain values).
PROTOTYPES = {
    "ddpm_conv2d_nhwc": [P, L, P, P, L, P, P, L, P, L] + [I] * 14 + [I, I, I, P, P, I, P],
    "ddpm_conv2d_wgrad_nhwc": [P, L, P, L, P, L] + [I] * 17 + [P],
    "ddpm_wgrad_effective_splits": [I, I, I],
    "ddpm_conv3x3_wgrad_splits": [I, I, I, I, I, I],
    "ddpm_conv3x3_wgrad_nhwc": [P, L, P, L, P, L, P, L, I, I, I, I, I, I, I, I, P],
    "ddpm_conv3x3_wgrad_up_nhwc": [P, L, P, L, P, L, P, L, I, I, I, I, I, I, I, I, P],
    "ddpm_conv1x1_wgrad_splits": [I, I, I],
    "ddpm_conv1x1_wgrad_nhwc": [P, L, P, L, P, L, P, L, I, I, I, I, I, P],
    "ddpm_wgrad_reduce": [P, I, P],
    "ddpm_wgrad_unpack": [P, P, P, I, F, P],
    "ddpm_wgrad_unpack_sumsq": [P, P, P, I, F, P, L, P],
    "ddpm_gemm": [P, L, L, I, P, L, L, I, P, L, L, P, P, L, L, I, I, I, I, F, I, I, I, I, P],
    "ddpm_groupnorm_silu_fwd": [P, L, P, L, P, P, P, P, I, I, I, I, F, I, F, U, P, I, P],
    "ddpm_groupnorm_silu_bwd": [P, L, P, L, P, L, P, P, P, P, P, P, I, I, I, I, I, F, U, P, I, P, L, P, L, I, P],
    "ddpm_gn_workspace_floats": [I, I, I, I, I],
    "ddpm_conv2d_variant": [L, L] + [I] * 18,
    "ddpm_conv2d_wgrad_variant": [L, L] + [I] * 17,
    "ddpm_gemm_variant": [L, I, L, I, L, I, I, I, I, I, I, I],
    "ddpm_attention_fwd": [P, L, P, L, I, I, I, F, I, P],
    "ddpm_attention_fwd_lse": [P, L, P, L, P, I, I, I, F, I, P],
    "ddpm_attention_bwd": [P, L, P, L, P, L, P, P, P, L, I, I, I, F, I, P],
    "ddpm_timestep_embedding": [P, P, P, I, I, P],
    "ddpm_nchw_to_nhwc": [P, P, I, I, I, I, I, P],
    "ddpm_pack_weight": [P, P, P, I, I, I, I, I, I, I, P],
    "ddpm_pack_weight_multi": [P, I, I, P],
    "ddpm_q_sample": [P, P, P, P, P, P, I, I, I, P],
    "ddpm_mse_fwd": [P, P, P, I, I, P],
    "ddpm_mse_bwd": [P, P, P, P, I, I, P],
    "ddpm_weighted_sum_f32": [P, P, P, I, P],
    "ddpm_atb_f32": [P, L, P, L, P, L, I, I, I, P],
    "ddpm_p_sample_step": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "ddpm_vlb_terms": [P] * 12 + [I, I, I, I, I, P],
    "ddpm_vlb_terms_bwd": [P] * 12 + [I, I, I, I, P],
    "ddpm_gather_i64": [P, P, P, I, P],
    "ddpm_add_i64": [P, I, L, P],
    "ddpm_gather_rows_f32": [P, P, P, I, I, I, P],
    "ddpm_silu_fwd": [P, P, L, P],
    "ddpm_silu_bwd": [P, P, P, L, I, P],
    "ddpm_colsum": [P, L, P, L, P, I, I, I, I, P],
    "ddpm_upsample2x_bwd": [P, P, L, I, I, I, I, I, I, P],
    "ddpm_resample2x_nhwc": [P, L, P, L, I, I, I, I, I, F, I, I, P],
    "ddpm_add_rows": [P, L, P, L, L, I, I, I, P],
    "ddpm_softmax_fwd": [P, P, L, I, I, P],
    "ddpm_softmax_bwd": [P, P, P, L, I, I, P],
    "ddpm_dropout_mask": [P, L, F, U, P],
    "ddpm_mfma_probe": [P, I, I, P],
    "ddpm_mt_sumsq_slots": [I],
    "ddpm_conv3x3_wgrad_variant": [I, I, I, I, I],
    "ddpm_wgrad3x3_ws_last_fault": [P],
    "ddpm_mt_grad_sumsq": [P, I, P, L, P],
    "ddpm_mt_adam_ema": [P, I, P, F, F, F, F, F, F, F, F, P, P],
    "ddpm_mt_gather_f32": [P, I, P],
    "ddpm_sumsq_accumulate": [P, L, P, P, P],
    "ddpm_adam_ema_step": [P, P, P, P, P, L, P, F, F, F, F, F, F, F, F, P],
    "ddpm_conv3x3_pc_last_fault": [P],
    "ddpm_set_reserved_cus": [I],
    "ddpm_get_reserved_cus": [],
    "ddpm_copy_probe": [P, P, L, I, P],
    # launch plans (csrc/plan.hip; driven by _plan.LaunchPlan)
    "ddpm_stream_order": [P, P],
    "ddpm_fill_zero": [P, L, P],
    "ddpm_plan_create": [],
    "ddpm_plan_destroy": [P],
    "ddpm_plan_append": [P, ctypes.c_char_p, P, I],
    "ddpm_plan_cut": [P],
    "ddpm_plan_segments": [P],
    "ddpm_plan_entries": [P],
    "ddpm_plan_run": [P, I],
    "ddpm_plan_failed_entry": [P, P],
    "ddpm_plan_entry_arity": [ctypes.c_char_p],
}
RESTYPES = {"ddpm_gn_workspace_floats": c_longlong, "ddpm_plan_create": c_void_p, "ddpm_plan_failed_entry": ctypes.c_char_p}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it was never built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"ddpm_torch (MI355X build): HIP library not found at {LIB_PATH}. "
                "Build it with `python ddpm-torch_amd/csrc/build.py` — there is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, c_int)
        _lib = handle
    return _lib


# ||g||^2 buffers (ddpm_mt_grad_sumsq / ddpm_wgrad_unpack_sumsq): a 64-float bank + one slot per block of the producing launch
# (64 per tensor row) — sized once for SUMSQ_MAX_TENSORS rows (2 MB); callers check their row count against it.
SUMSQ_MAX_TENSORS = 8192
SUMSQ_FLOATS = 64 + 64 * SUMSQ_MAX_TENSORS


def _invoke(name, args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")


_recorder = None        # a _plan.LaunchPlan while a step is being recorded: every call is executed AND appended to it


def call(name, *args):
    """Enqueue one launching entry point (raises on a non-zero status).  While a launch plan records, the call is also appended to it."""
    if _recorder is not None:
        _recorder.add(name, args)
    _invoke(name, args)


def record_into(plan):
    """Route a copy of every call() to ``plan`` (None: stop).  Returns the previous recorder."""
    global _recorder
    prev, _recorder = _recorder, plan
    return prev


def retain(t):
    """Called by the engine for every tensor it allocates: while a plan records, the plan keeps them alive — its entries address them
    by raw pointer (host-emulated runs; on the GPU the recording also runs inside a private allocator pool, see _plan.py)."""
    if _recorder is not None and _recorder.retains:
        _recorder.keep.append(t)
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_routed = None


def route_stream(handle):
    """Make stream() return `handle` (a raw hipStream_t) until route_stream(previous) — for code that launches a few kernels on
    another stream without switching torch's current stream.  Returns the previous routing."""
    global _routed
    prev, _routed = _routed, handle
    return prev


def stream():
    """hipStream_t the next launch goes to: the routed one if any, else torch's current stream (the raw getter skips building a
    torch.cuda.Stream per launch)."""
    if _routed is not None:
        return _routed
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ddpm_torch (MI355X build) runs on CUDA/HIP tensors only; "
                               "got a CPU tensor and there is no CPU fallback")


def on_device(t):
    """True when the C-ABI kernels can address ``t`` (a CUDA/HIP tensor)."""
    return t.is_cuda


def ptr(t):
    return 0 if t is None else t.data_ptr()
