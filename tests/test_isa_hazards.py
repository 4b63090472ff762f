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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_conv3x3_fragment_registers_are_not_touched_before_their_counted_waits(tmp_path):
    """csrc/conv3x3.hip: the wave-specialised 3x3 kernel (and its two predecessors, which still serve the residual / 8x8 calls) read their MFMA
    fragments with inline-asm ds_read_b128 and retire them with COUNTED `s_waitcnt lgkmcnt(n)` statements that are not tied to the destination
    registers (a "+v" tie makes hipcc copy the registers while the data is in flight).  Correctness therefore rests on what the register
    allocator places between a read and its wait — at 241 of 256 VGPRs.  tests/isa_lgkm.py runs a flag-partitioned forward data flow over
    the generated code: no instruction (MFMA, copy, accumulator move, spill) may touch a VGPR while an LDS read into it is outstanding; and the
    kernels must not spill.  The checker is itself checked by mutation: relaxing any single counted wait by one must be reported (a fifth of
    the waits are sampled; redundant waits — the first sub-step of a tile, a wait dominated by a stricter one — account for the misses)."""
    from tests.isa_lgkm import check_lgkm, kernel_bodies
    out = tmp_path / "c3.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "ddpm-torch_amd", "csrc", "conv3x3.hip")], check=True, capture_output=True)
    text = out.read_text()
    bodies = dict(kernel_bodies(text))
    assert len(bodies) == 3 and sum("conv3x3_pc_kernel" in k for k in bodies) == 1
    for name, body in bodies.items():
        violations, stats = check_lgkm(body)
        assert not violations, name + "\n" + "\n".join(violations[:10])
        assert stats["scratch"] == 0 and stats["mfma"] > 30 and stats["lds_reads"] > 60, (name, stats)
    for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", text):
        assert int(m.group(1)) == 0
    assert len(re.findall(r"\.vgpr_spill_count:", text)) == 3
    pc = next(b for k, b in bodies.items() if "conv3x3_pc_kernel" in k)
    waits = [i for i, l in enumerate(pc) if re.search(r"s_waitcnt lgkmcnt\([1-9]\)", l)]
    assert len(waits) > 150                                   # the counted waits of the unrolled tap loop
    sample, caught = waits[::5], 0
    for i in sample:
        mutated = list(pc)
        k = int(re.search(r"lgkmcnt\((\d+)\)", mutated[i]).group(1))
        mutated[i] = re.sub(r"lgkmcnt\(\d+\)", f"lgkmcnt({k + 1})", mutated[i])
        caught += bool(check_lgkm(mutated)[0])
    assert caught >= 0.75 * len(sample), (caught, len(sample))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("source,kernels", [("wgrad.hip", 4), ("wgrad1x1.hip", 1), ("pointwise.hip", 3)])
def test_other_asm_read_kernels_hold_the_same_property(tmp_path, source, kernels):
    """The weight-gradient and 1x1 kernels read their fragments the same way (inline-asm ds_read / ds_read_b64_tr_b16 with hand-placed
    waits): same data-flow check.  Not covered: csrc/attention.hip — its fragment pre-reads sit in front of loops whose trip count the
    compiler cannot prove positive (`nks >= 1` follows from `ch * KR < Lpad`), and on the zero-trip edge, which no launch can take, the
    registers are reused without a wait; the checker cannot rule that edge out and reports it."""
    from tests.isa_lgkm import check_lgkm, kernel_bodies
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "ddpm-torch_amd", "csrc", source)], check=True, capture_output=True)
    bodies = dict(kernel_bodies(out.read_text()))
    assert len(bodies) == kernels
    for name, body in bodies.items():
        violations, stats = check_lgkm(body)
        assert not violations, name + "\n" + "\n".join(violations[:10])
        assert stats["lds_reads"] >= 20 and stats["mfma"] >= 8, (name, stats)
