"""Developer tool: print a window of a rocprofv3 --kernel-trace CSV as a timeline (start offset, duration, gap to the previous
kernel end on the same queue) -- used to see what a residual check costs the sweep stream.
usage: python tools/timeline.py <kernel_trace.csv> [first_row_fraction=0.5] [rows=60]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
i0 = int(len(rows) * frac)
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
for r in rows[i0:i0 + n]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:7.2f}  gap {gap:6.2f}  q{q:>3s}  grid {r.get('Grid_Size_X', '?'):>7s}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}  {r['Kernel_Name'][:70]}")
