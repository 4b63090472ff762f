"""CPU: the C-ABI shared object loads (no GPU needed) and exports exactly the entry points include/*.h declare (ddpm_hip.h = the
interface that replaces the reference's ATen calls; ddpm_hip_debug.h = instrumentation and test hooks, none of which the product path needs);
the ctypes table binds every one of them with the right arity."""
import ctypes
import os
import re

import pytest

from ddpm_torch import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ddpm_hip.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "ddpm_hip_debug.h")          # instrumentation / measurement / test hooks: a separate header


def declared(paths=(HEADER, DEBUG_HEADER)):
    src = "".join(re.sub(r"/\*.*?\*/", "", open(p).read(), flags=re.S) for p in paths)
    out = {}
    for m in re.finditer(r"\b(?:int|long long|void\s*\*|const char\s*\*)\s*(ddpm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = len(args)
    return out


@pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")
def test_library_exports_every_declared_symbol():
    decl = declared()
    assert len(decl) >= 25
    handle = ctypes.CDLL(_hip.LIB_PATH)
    for name in decl:
        assert hasattr(handle, name), f"{name} declared in ddpm_hip.h but not exported"
    assert set(decl) == set(_hip.PROTOTYPES), set(decl) ^ set(_hip.PROTOTYPES)
    for name, nargs in decl.items():
        assert len(_hip.PROTOTYPES[name]) == nargs, (name, nargs, len(_hip.PROTOTYPES[name]))
    _hip.lib()
    # the interface header carries no instrumentation: every hook lives in the debug header
    hooks = {"ddpm_conv2d_variant", "ddpm_conv2d_wgrad_variant", "ddpm_gemm_variant", "ddpm_mfma_probe", "ddpm_copy_probe", "ddpm_dropout_mask",
             "ddpm_conv3x3_pc_last_fault", "ddpm_set_reserved_cus", "ddpm_get_reserved_cus"}
    assert hooks <= set(declared((DEBUG_HEADER,))) and not hooks & set(declared((HEADER,)))


@pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")
def test_argument_validation_returns_status_codes_without_a_gpu():
    lib = _hip.lib()
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.ddpm_q_sample(0, 0, 0, 0, 0, 0, 1, 1, 10, 0) == 5
    assert lib.ddpm_gn_workspace_floats(2, 64, 100, 32, 1) == -1          # 100 channels not divisible by 32
    assert lib.ddpm_gn_workspace_floats(2, 64, 128, 32, 1) > 0
    with pytest.raises(RuntimeError, match="null pointer"):
        _hip.call("ddpm_silu_fwd", 0, 0, 10, 0)
    # maximum sizes: the kernels address an operand with 32-bit byte offsets, so a tensor of 2 GiB or more is refused on the host
    # (32 x 32 x 384 channels bf16: B = 4096 is 3 GiB; B = 2048 = 1.5 GiB passes validation and only fails here for want of a GPU)
    fake = 0x10000
    conv = lambda B: lib.ddpm_conv2d_nhwc(fake, 384, fake, fake, 128, 0, 0, 0, 0, 0, B, 32, 32, 384, 32, 32, 128, 3, 3, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 1, 0)
    assert conv(4096) == 1 and conv(2048) in (0, 4)


QUERIES = {"ddpm_wgrad_effective_splits", "ddpm_conv3x3_wgrad_splits", "ddpm_conv1x1_wgrad_splits", "ddpm_gn_workspace_floats",
           "ddpm_conv2d_variant", "ddpm_conv2d_wgrad_variant", "ddpm_gemm_variant", "ddpm_conv3x3_pc_last_fault", "ddpm_set_reserved_cus",
           "ddpm_get_reserved_cus", "ddpm_mt_sumsq_slots", "ddpm_conv3x3_wgrad_variant", "ddpm_wgrad3x3_ws_last_fault"}


@pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")
def test_plan_executor_knows_every_launching_entry_point_with_the_declared_arity():
    """csrc/plan.hip generates its call thunks from the header's declarations; the ctypes table is what the recorder converts arguments
    with — the two must agree for every entry point a step can record."""
    lib = _hip.lib()
    for name, argtypes in _hip.PROTOTYPES.items():
        arity = lib.ddpm_plan_entry_arity(name.encode())
        if name in QUERIES or name.startswith("ddpm_plan_"):
            assert arity == -1, name                               # pure queries / the plan API itself enqueue nothing: not recordable
        else:
            assert arity == len(argtypes), (name, arity, len(argtypes))


@pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")
def test_plan_api_validates_on_the_host():
    lib = _hip.lib()
    h = ctypes.c_void_p(lib.ddpm_plan_create())
    assert h.value
    words = (ctypes.c_ulonglong * 4)(0, 0, 10, 0)
    as_ptr = ctypes.cast(words, ctypes.c_void_p)
    assert lib.ddpm_plan_append(h, b"ddpm_silu_fwd", as_ptr, 4) == 0            # first entry
    assert lib.ddpm_plan_append(h, b"ddpm_silu_fwd", as_ptr, 3) == -1           # wrong argument count
    assert lib.ddpm_plan_append(h, b"ddpm_no_such_entry", as_ptr, 4) == -1
    assert lib.ddpm_plan_append(h, b"ddpm_gemm_variant", as_ptr, 4) == -1       # a query is not a launch
    assert lib.ddpm_plan_entries(h) == 1 and lib.ddpm_plan_segments(h) == 1
    assert lib.ddpm_plan_cut(h) == 0 and lib.ddpm_plan_segments(h) == 1
    assert lib.ddpm_plan_append(h, b"ddpm_silu_fwd", as_ptr, 4) == 1 and lib.ddpm_plan_segments(h) == 2
    # running it calls ddpm_silu_fwd(NULL, NULL, 10, stream 0): rejected on the host (null pointer = 5) before any launch, and reported
    assert lib.ddpm_plan_run(h, 0) == 5
    idx = ctypes.c_int(-1)
    name = lib.ddpm_plan_failed_entry(h, ctypes.cast(ctypes.pointer(idx), ctypes.c_void_p))
    assert name == b"ddpm_silu_fwd" and idx.value == 0
    assert lib.ddpm_plan_run(h, 7) == 1                                         # no such segment
    assert lib.ddpm_fill_zero(0, 16, 0) == 5 and lib.ddpm_fill_zero(0, 0, 0) == 0 and lib.ddpm_fill_zero(0, -1, 0) == 1
    assert lib.ddpm_plan_destroy(h) == 0
