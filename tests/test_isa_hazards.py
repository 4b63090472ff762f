"""Not a GPU test: compiles csrc/pointwise.hip to gfx950 ISA and checks a property the C++ source cannot express.

The 1x1 kernel requests its residual / "+=" rows with INLINE-ASM loads and waits for them with a hand-placed `s_waitcnt vmcnt(N)` a K-step
later (a C++ load would make hipcc drain every pending LDS-DMA first: csrc/pointwise.hip).  The compiler believes an asm output is valid
right behind the asm statement, so a register copy, a spill or a reuse scheduled between the load and the wait would read or clobber
registers that are still in flight — correct today only by register allocation.  This test makes that a checked property:
in every instantiation, no instruction between an asm load and the next hand-placed vmcnt wait touches the load's destination."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def kernels(asm):
    lines = asm.split("\n")
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l):
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            yield l.split(":")[0], lines[i:end]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_pointwise_asm_loads_are_not_touched_before_their_wait(tmp_path):
    out = tmp_path / "pw.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "ddpm-torch_amd", "csrc", "pointwise.hip")], check=True, capture_output=True)
    seen = 0
    for name, body in kernels(out.read_text()):
        if "pw_conv_kernel" not in name:
            continue
        loads = 0
        for i, l in enumerate(body):
            m = re.search(r"global_load_dwordx2 v\[(\d+):(\d+)\]", l)
            if not m or ";;#ASMSTART" not in body[i - 1]:
                continue
            loads += 1
            dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for k in body[i + 1:]:
                k = k.strip()
                if not k or k[0] in ";.":
                    continue
                if re.match(r"s_waitcnt vmcnt\(\d+\)", k):
                    break                                   # the hand-placed wait (the only vmcnt waits in this kernel are ours)
                if "global_load_dwordx2" in k:
                    continue
                used = {int(a) for a in re.findall(r"\bv(\d+)\b", k)}
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", k):
                    used |= set(range(int(a), int(b) + 1))
                assert not (used & dest), f"{name}: `{k}` touches the destination of `{l.strip()}` before its wait"
        assert loads in (16, 32), (name, loads)             # 8 MJ residual + 8 MJ "+=" requests per lane
        seen += 1
    assert seen == 3                                        # <128,256>, <256,128>, <128,128>
