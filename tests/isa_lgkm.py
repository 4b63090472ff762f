"""TEST INFRASTRUCTURE: a forward data-flow check over gfx950 assembly — is any VGPR touched while an LDS read into it is still in flight?

The wave-specialised kernels fill their fragment registers with inline-asm ``ds_read_b128`` and retire them with separate, COUNTED
``s_waitcnt lgkmcnt(n)`` statements (csrc/conv3x3.hip).  The compiler believes an asm output is valid right behind the asm statement: a
copy, a spill or an accumulator move that register allocation places between a read and the wait that covers it would use data still in
flight, and nothing in the C++ source can forbid it.  This module makes it a checked property of the generated code.

Model (one wave): LDS operations complete in issue order; ``s_waitcnt lgkmcnt(n)`` returns when at most n are outstanding, so a read is
complete iff at least n LDS operations were issued after it.  Abstract state per program point: {vgpr -> fewest LDS operations issued
since the pending read into it, over all paths reaching the point}; joins take the union with the minimum (the conservative side).

hipcc lowers the kernels' uniform booleans (``pend``, ``tile_end`` ...) into SGPR-pair flags and re-tests them at merged blocks
(``s_mov_b64 s[4:5], -1`` ... ``s_and_b64 vcc, exec, s[4:5]`` / ``s_cbranch_vccz``), so a path-insensitive analysis walks paths the
program cannot take and reports reads "in flight" across them.  The analysis is therefore partitioned by the KNOWN values of those flags:
constants moved into SGPR pairs, their copies, the ``v_cndmask 0,1`` / ``v_cmp_ne 1`` inversion idiom, and vcc derived from them; a
conditional branch on a known vcc follows only the feasible edge.  Anything it cannot evaluate is unknown and both edges are followed
(false alarms possible, missed hazards not).  A violation = an instruction that reads or writes a VGPR of the state (other than an LDS
read re-targeting it).  Scalar memory loads share the counter but may return out of order: a counted wait (n > 0) with one outstanding is
reported as well.
"""
import re

_LDS = re.compile(r"^(ds_\w+)\b")
_VREG = re.compile(r"\bv(\d+)\b")
_VRANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
_SPAIR = re.compile(r"^s\[(\d+):(\d+)\]$")
_SREG = re.compile(r"^s(\d+)$")


def kernel_bodies(asm_text):
    lines = asm_text.split("\n")
    functions = set(re.findall(r"^\s*\.type\s+(_Z\w+),@function", asm_text, flags=re.M))      # (data symbols have labels too)
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and l.split(":")[0] in functions:
            end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(".amdhsa_kernel") or lines[j].startswith(".Lfunc_end"))
            yield l.split(":")[0], lines[i + 1:end]


def _vregs(text):
    used = {int(a) for a in _VREG.findall(text)}
    for a, b in _VRANGE.findall(text):
        used |= set(range(int(a), int(b) + 1))
    return used


def _instructions(body):
    """[(label or None, instruction text)] with comments and directives stripped."""
    out, pending_label = [], None
    for raw in body:
        s = raw.strip()
        m = _LABEL.match(s)
        if m:
            pending_label = m.group(1)
            continue
        line = "" if s.startswith(";") else raw.split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        out.append((pending_label, line))
        pending_label = None
    return out


def _operands(text):
    parts = text.split(None, 1)
    return parts[0], ([o.strip() for o in parts[1].split(",")] if len(parts) > 1 else [])


def _sgprs(op):
    m = _SPAIR.match(op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = _SREG.match(op)
    return {int(m.group(1))} if m else set()


_NO_DEST = ("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_branch", "s_cbranch", "s_endpgm", "s_setprio", "s_sethalt",
            "s_trap", "s_sendmsg", "s_setreg", "s_icache", "s_dcache", "global_store", "buffer_store", "ds_write", "ds_add_u32", "flat_store",
            "scratch_store", "s_store", "s_memtime_dummy")


def _flag_transfer(flags, mnem, ops, text):
    """Update the known-flag map (keys: ('s', lo, hi) | 'vcc' | ('v', n); values 0 / 1) for one instruction."""
    def kill_s(regs):
        for k in [k for k in flags if isinstance(k, tuple) and k[0] == "s" and regs & set(range(k[1], k[2] + 1))]:
            del flags[k]

    def key_of(op):
        m = _SPAIR.match(op)
        return ("s", int(m.group(1)), int(m.group(2))) if m else ("vcc" if op == "vcc" else None)

    def val_of(op):
        if op in ("0",):
            return 0
        if op in ("-1", "exec"):
            return 1
        k = key_of(op)
        return flags.get(k) if k is not None else None

    if mnem.startswith(_NO_DEST):
        return
    dest = ops[0] if ops else ""
    dk = key_of(dest)
    new = None
    if mnem == "s_mov_b64" and len(ops) == 2:
        new = val_of(ops[1])
    elif mnem in ("s_and_b64", "s_andn2_b64") and len(ops) == 3 and "exec" in ops[1:]:
        other = ops[2] if ops[1] == "exec" else ops[1]
        v = val_of(other)
        if v is not None:
            new = v if mnem == "s_and_b64" else (1 - v if ops[1] == "exec" else None)
    elif mnem == "v_cndmask_b32_e64" and len(ops) == 4 and ops[1] == "0" and ops[2] == "1" and re.match(r"^v\d+$", dest):
        v = val_of(ops[3])
        n = int(dest[1:])
        if v is None:
            flags.pop(("v", n), None)
        else:
            flags[("v", n)] = v
        return
    elif mnem in ("v_cmp_ne_u32_e64", "v_cmp_eq_u32_e64") and len(ops) == 3 and ops[1] == "1" and re.match(r"^v\d+$", ops[2]):
        v = flags.get(("v", int(ops[2][1:])))
        if v is not None:
            new = (1 - v) if mnem == "v_cmp_ne_u32_e64" else v
    # every other instruction (and the ones above when their source is unknown): the destination becomes unknown
    if dk == "vcc":
        flags.pop("vcc", None)
    regs = _sgprs(dest)
    if regs:
        kill_s(regs)
    if dest.startswith("v"):
        for n in _vregs(dest):
            flags.pop(("v", n), None)
    # implicit writers of vcc
    if mnem.startswith("v_cmp") and not mnem.endswith("_e64") or "_co_" in mnem or mnem.startswith(("v_addc", "v_subb", "v_div_scale", "v_cmpx")):
        flags.pop("vcc", None)
    if "vcc" in ops[1:2] and mnem.startswith(("v_add_co", "v_sub_co", "v_mad_u64", "v_mad_i64")):
        flags.pop("vcc", None)
    if new is not None and dk is not None:
        flags[dk] = new


def check_lgkm(body, max_states=400000):
    """Returns (violations, stats).  violations: list of human-readable strings (empty = the property holds)."""
    ins = _instructions(body)
    n = len(ins)
    parsed = [_operands(t) for _, t in ins]
    label_at = {lab: i for i, (lab, _) in enumerate(ins) if lab}
    # flags worth tracking: SGPR pairs that feed a branch condition (directly, through a copy, or through the inversion idiom)
    violations = {}
    smem_reports = {}

    def step(i, pending, smem, flags):
        """Apply instruction i to the abstract state; returns list of (next index, ...) successors is done by the caller."""
        text = ins[i][1]
        mnem, ops = parsed[i]
        m = _LDS.match(text)
        if text.startswith("s_waitcnt"):
            c = re.search(r"lgkmcnt\((\d+)\)", text)
            if c:
                k = int(c.group(1))
                if k > 0 and smem:
                    smem_reports[i] = f"[{i}] `{text}`: counted lgkmcnt wait with a scalar memory load outstanding (SMEM returns out of order)"
                if k == 0:
                    smem = 0
                pending = {r: y for r, y in pending.items() if y < k}
            return pending, smem
        if text.startswith(("s_load_", "s_buffer_load_")):
            _flag_transfer(flags, mnem, ops, text)
            return pending, smem + 1
        regs = _vregs(text)
        if m:
            is_read = any(w in m.group(1) for w in ("read", "_rtn", "permute", "swizzle"))
            dest = _vregs(ops[0]) if is_read and ops else set()
            clash = (regs - dest) & set(pending)
            if clash:
                violations[i] = f"[{i}] `{text}` uses v{sorted(clash)} while an LDS read into it is in flight"
            pending = {r: y + 1 for r, y in pending.items()}
            for r in dest:
                pending[r] = 0
                flags.pop(("v", r), None)
            return pending, smem
        clash = regs & set(pending)
        if clash:
            violations[i] = (f"[{i}] `{text}` touches v{sorted(clash)} before the lgkmcnt wait that covers the LDS read into it "
                             f"(LDS operations issued since: {[pending[r] for r in sorted(clash)]})")
        _flag_transfer(flags, mnem, ops, text)
        return pending, smem

    # worklist over (instruction index at a block entry, known flags); states with the same key merge by minimum
    states = {}
    work = [(0, frozenset())]
    states[(0, frozenset())] = ({}, 0)
    visited = 0
    while work:
        key = work.pop()
        i, fl = key
        pending, smem = states[key]
        pending, flags = dict(pending), dict(fl)
        while True:
            visited += 1
            if visited > max_states * 50:
                raise AssertionError("state explosion in the lgkmcnt data flow")
            text = ins[i][1]
            pending, smem = step(i, pending, smem, flags)
            m = _BRANCH.match(text)
            nxt = []
            if m:
                tgt = label_at.get(m.group(2))
                kind = m.group(1)
                take = fall = True
                if kind == "s_branch":
                    fall = False
                elif kind in ("s_cbranch_vccz", "s_cbranch_vccnz") and "vcc" in flags:
                    zero = flags["vcc"] == 0
                    take = zero if kind == "s_cbranch_vccz" else not zero
                    fall = not take
                if take and tgt is not None:
                    nxt.append(tgt)
                if fall and i + 1 < n:
                    nxt.append(i + 1)
            elif text.startswith("s_endpgm"):
                nxt = []
            elif i + 1 < n and ins[i + 1][0] is None:
                i += 1
                continue                                     # straight-line code: stay in this run
            elif i + 1 < n:
                nxt = [i + 1]
            for j in nxt:
                k2 = (j, frozenset(flags.items()))
                cur = states.get(k2)
                if cur is None:
                    states[k2] = (dict(pending), smem)
                    work.append(k2)
                else:
                    merged, changed = dict(cur[0]), False
                    for r, y in pending.items():
                        if r not in merged or y < merged[r]:
                            merged[r] = y
                            changed = True
                    sm = max(cur[1], smem)
                    if changed or sm != cur[1]:
                        states[k2] = (merged, sm)
                        work.append(k2)
            break
        if len(states) > max_states:
            raise AssertionError("state explosion in the lgkmcnt data flow")
    out = [violations[i] for i in sorted(violations)] + [smem_reports[i] for i in sorted(smem_reports)]
    stats = dict(instructions=n, states=len(states),
                 lds_reads=sum(1 for _, t in ins if t.startswith("ds_read")),
                 counted_waits=sum(1 for _, t in ins if re.search(r"lgkmcnt\(([1-9]\d*)\)", t)),
                 mfma=sum(1 for _, t in ins if t.startswith("v_mfma")),
                 scratch=sum(1 for _, t in ins if t.startswith("scratch_")))
    return out, stats
