"""Developer tool: where the sweep queue idles.  Reads a rocprofv3 --kernel-trace CSV, takes the k_sweep_q launches in start
order and bins the gaps between one launch's end and the next one's start.
usage: python tools/gap_hist.py <kernel_trace.csv> [skip_first_fraction=0.3]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_sweep_q" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)):]
bins = [(0, 0.5), (0.5, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 1e9)]
cnt = [0] * len(bins); tot = [0.0] * len(bins)
busy = 0.0
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    busy += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    for i, (lo, hi) in enumerate(bins):
        if lo <= gap < hi:
            cnt[i] += 1; tot[i] += gap
n = len(rows) - 1
span = (int(rows[-1]["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{n} launches, span {span / 1e3:.2f} ms, kernel time {busy / 1e3:.2f} ms ({busy / n:.2f} us each), idle {(span - busy) / 1e3:.2f} ms ({(span - busy) / n:.2f} us per launch)")
for (lo, hi), c, t in zip(bins, cnt, tot):
    print(f"  gap {lo:5.1f} .. {hi if hi < 1e8 else float('inf'):5.1f} us: {c:6d} gaps, {t / 1e3:7.2f} ms in all, {t / max(n, 1):6.3f} us per launch")
