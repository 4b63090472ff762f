"""Per-step GPU timeline from a rocprofv3 kernel trace (csv): for the steps between consecutive optimiser launches, busy time per
HSA queue, union-busy / idle time of the device, and per-kernel totals by queue.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py ...
    python scripts/timeline.py DIR [marker-kernel-substring]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main(d, marker="mt_adam_ema_kernel"):
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        print("not enough steps", len(rows), len(marks)); return
    a, b = marks[-3], marks[-2]          # one full step: after optimiser launch a up to and including optimiser launch b
    step = rows[a + 1:b + 1]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print(f"step wall {1e-6 * (t1 - t0):.3f} ms, {len(step)} dispatches")
    byq = defaultdict(list)
    for s, e, n, q in step:
        byq[q].append((s, e, n))
    def union(iv):
        iv = sorted(iv); tot = 0; cs, ce = iv[0]
        for s, e in iv[1:]:
            if s > ce: tot += ce - cs; cs, ce = s, e
            else: ce = max(ce, e)
        return tot + ce - cs
    print(f"device busy (union) {1e-6 * union([(s, e) for s, e, _, _ in step]):.3f} ms")
    for q, ks in byq.items():
        busy = union([(s, e) for s, e, _ in ks])
        print(f"-- queue {q}: {len(ks)} dispatches, busy {1e-6 * busy:.3f} ms, sum {1e-6 * sum(e - s for s, e, _ in ks):.3f} ms")
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in ks:
            agg[short(n)][0] += 1; agg[short(n)][1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
            print(f"   {1e-6 * t:8.3f} ms  n={c:4d}  avg {1e-3 * t / c:7.1f} us  {n}")
    # gaps on the busiest queue
    q = max(byq, key=lambda k: len(byq[k]))
    ks = sorted(byq[q])
    gaps = sorted(((ks[i + 1][0] - ks[i][1], short(ks[i][2]), short(ks[i + 1][2])) for i in range(len(ks) - 1)), reverse=True)
    print("largest gaps on queue", q, "(us, after, before):")
    for g, x, y in gaps[:12]:
        print(f"   {1e-3 * g:7.1f}  {x}  ->  {y}")
    print(f"   total gap {1e-6 * sum(g for g, _, _ in gaps if g > 0):.3f} ms over {len(gaps)} boundaries")
    # the tail of the step: what runs (and what waits) between the end of the backward's critical path and the optimiser
    print("last 28 dispatches (start, end in us before the end of the step; queue; kernel):")
    for s_, e_, n, q_ in sorted(step, key=lambda r: r[0])[-28:]:
        print(f"   {1e-3 * (t1 - s_):8.1f} {1e-3 * (t1 - e_):8.1f}  q{q_}  {short(n) or '(anonymous-namespace kernel)'}")


if __name__ == "__main__":
    main(*sys.argv[1:])
