"""Developer tool: per queue of a rocprofv3 --kernel-trace CSV, the gaps between consecutive sweep kernels and the share of the
span during which NO sweep kernel of any queue ran.  usage: python tools/stream_gaps.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_sweep" in r["Kernel_Name"]]
q = defaultdict(list)
for r in rows:
    q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "z" if "k_sweep_z" in r["Kernel_Name"] else "q"))
for k, v in q.items():
    v.sort()
    v = v[len(v) // 4:]
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(v, v[1:])]
    dur = {t: [(e - s) / 1e3 for s, e, tt in v if tt == t] for t in ("z", "q")}
    span = (v[-1][1] - v[0][0]) / 1e3
    print(f"queue {k}: {len(v)} sweep kernels over {span / 1e3:.2f} ms; z {len(dur['z'])} x {sum(dur['z']) / max(len(dur['z']), 1):.1f} us, q {len(dur['q'])} x {sum(dur['q']) / max(len(dur['q']), 1):.1f} us; "
          f"gaps: mean {sum(gaps) / len(gaps):.2f} us, sum {sum(gaps) / 1e3:.2f} ms = {100 * sum(gaps) / span:.1f} % of the span; > 10 us: {sum(1 for g in gaps if g > 10)}")
ev = sorted((t, d) for s, e, _ in sum(q.values(), []) for t, d in ((s, 1), (e, -1)))
t0 = ev[len(ev) // 4][0]                                  # (the first quarter -- warm-up, setup -- is not counted)
depth, last, idle, one = 0, ev[0][0], 0, 0
for t, d in ev:
    if t > t0:
        if depth == 0: idle += t - max(last, t0)
        if depth == 1: one += t - max(last, t0)
    depth += d; last = t
span = ev[-1][0] - t0
print(f"all queues: no sweep kernel running {100 * idle / span:.1f} % of the span, exactly one {100 * one / span:.1f} %")
