"""CPU: the C-ABI shared object loads (no GPU needed) and exports exactly the entry points include/ddpm_hip.h declares;
the ctypes table binds every one of them with the right arity."""
import ctypes
import os
import re

import pytest

from ddpm_torch import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ddpm_hip.h")


def declared():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|long long)\s+(ddpm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
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


@pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH), reason="libddpm_hip.so not built")
def test_argument_validation_returns_status_codes_without_a_gpu():
    lib = _hip.lib()
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.ddpm_q_sample(0, 0, 0, 0, 0, 0, 1, 1, 10, 0) == 5
    assert lib.ddpm_gn_workspace_floats(2, 64, 100, 32, 1) == -1          # 100 channels not divisible by 32
    assert lib.ddpm_gn_workspace_floats(2, 64, 128, 32, 1) > 0
    with pytest.raises(RuntimeError, match="null pointer"):
        _hip.call("ddpm_silu_fwd", 0, 0, 10, 0)
