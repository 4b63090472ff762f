"""Where does a training step's wall time go?  From a rocprofv3 --kernel-trace directory: per steady-state step, the span, the time
with >= 1 kernel running (union over both streams), the idle time, per-queue busy time and the biggest idle gaps with the kernels
around them.   python scripts/step_timeline.py DIR [first_step last_step]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "")) for r in rows))
# step boundaries: the optimiser kernel ends a step
ends = [e for s, e, k, q, st in ev if "mt_adam_ema" in k]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 6
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(ends) - 1
t0, t1 = ends[lo], ends[hi]
win = [(s, e, k, q, st) for s, e, k, q, st in ev if s >= t0 and e <= t1 + 200000]
nsteps = hi - lo
span = (t1 - t0) / nsteps / 1e6
busy = collections.defaultdict(float)
for s, e, k, q, st in win:
    busy[(q, st)] += e - s
# union
cur_s, cur_e, union, gaps = None, None, 0, []
last_k = None
for s, e, k, q, st in sorted(win):
    if cur_e is None:
        cur_s, cur_e, last_k = s, e, k
    elif s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, last_k, k))
        cur_s, cur_e, last_k = s, e, k
    else:
        if e > cur_e:
            cur_e, last_k = e, k
union += cur_e - cur_s
print(f"steps {lo}..{hi}: span {span:.3f} ms/step; some kernel running {union / nsteps / 1e6:.3f} ms/step; idle {span - union / nsteps / 1e6:.3f} ms/step; launches/step {len(win) / nsteps:.0f}")
for (q, st), b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q} stream {st}: busy {b / nsteps / 1e6:.3f} ms/step")
tot_gap = sum(g for g, _, _ in gaps)
print(f"idle gaps: {len(gaps) / nsteps:.0f}/step, total {tot_gap / nsteps / 1e6:.3f} ms/step; > 5 us: {sum(g for g, _, _ in gaps if g > 5000) / nsteps / 1e6:.3f} ms/step")
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    key = (a.split("(")[0][-60:], b.split("(")[0][-60:])
    agg[key][0] += 1; agg[key][1] += g
for (a, b), (n, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {g / nsteps / 1e3:7.1f} us/step  n/step={n / nsteps:5.1f}  after [{a}] before [{b}]")

# per-queue kernel breakdown (what sits on the critical stream?)
perq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for s_, e_, k, q, st in win:
    name = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
    perq[(q, st)][name][0] += 1; perq[(q, st)][name][1] += e_ - s_
for key, d in perq.items():
    print(f"--- queue {key[0]} stream {key[1]}: {sum(v[0] for v in d.values()) / nsteps:.0f} launches/step")
    for name, (n, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {t / nsteps / 1e3:8.1f} us/step  n/step={n / nsteps:5.1f}  avg {t / n / 1e3:6.1f} us  {name}")

# gaps of the BUSIEST queue (the critical stream) while it waits — for the other stream, or for the host: where does the step's span exceed
# that stream's kernel time?
main = max(busy.items(), key=lambda kv: kv[1])[0]
mk = sorted((s_, e_, k) for s_, e_, k, q, st in win if (q, st) == main)
mg = collections.defaultdict(lambda: [0, 0])
tot = 0
for (s0, e0, k0), (s1, e1, k1) in zip(mk, mk[1:]):
    if s1 - e0 > 3000:
        short = lambda k: k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[-48:]
        mg[(short(k0), short(k1))][0] += 1; mg[(short(k0), short(k1))][1] += s1 - e0
        tot += s1 - e0
print(f"--- gaps > 3 us on queue {main[0]} stream {main[1]} (the critical stream): {tot / nsteps / 1e3:.1f} us/step")
for (a, b), (n_, g) in sorted(mg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {g / nsteps / 1e3:7.1f} us/step  n/step={n_ / nsteps:5.1f}  after [{a}] before [{b}]")

# TL_TAIL=N: the last N kernels of one step on both streams (start / end in us before the step's optimiser kernel ends): what the join waits for
import os
if os.environ.get("TL_TAIL"):
    n_tail = int(os.environ["TL_TAIL"])
    t_end = ends[hi]
    one = [(s_, e_, k, q, st) for s_, e_, k, q, st in ev if ends[hi - 1] < s_ and e_ <= t_end]
    print(f"--- tail of step {hi}: last {n_tail} kernels (us before the end of the step)")
    for s_, e_, k, q, st in sorted(one)[-n_tail:]:
        name = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        print(f"  q{q}  start -{(t_end - s_) / 1e3:8.1f}  end -{(t_end - e_) / 1e3:8.1f}  dur {(e_ - s_) / 1e3:7.1f}  {name}")
if os.environ.get("TL_HEAD"):
    n_head = int(os.environ["TL_HEAD"])
    t_beg = ends[hi - 1]
    one = [(s_, e_, k, q, st) for s_, e_, k, q, st in ev if t_beg <= s_ and e_ <= ends[hi]]
    print(f"--- head of step {hi}: first {n_head} kernels (us after the previous step's optimiser kernel ended)")
    for s_, e_, k, q, st in sorted(one)[:n_head]:
        name = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        print(f"  q{q}  start +{(s_ - t_beg) / 1e3:8.1f}  end +{(e_ - t_beg) / 1e3:8.1f}  dur {(e_ - s_) / 1e3:7.1f}  {name}")
