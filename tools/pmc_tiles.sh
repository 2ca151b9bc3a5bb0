#!/bin/bash
# SQ counters of the blocked sweep per tile shape (option "tile" = waves*100 + rows per lane).
# Usage: tools/pmc_tiles.sh <tag> <workload> <tile> [<tile> ...]     -> gpurun_out/<tag>/tiles_<workload>.txt
set -e
tag=$1; wl=$2; shift; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
for t in "$@"; do
  d=$out/tiles/$wl/$t; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $d -- python3 $root/tools/pmc_workload.py $wl 200 tile=$t > $d/log.txt 2>&1 || echo "pass failed for tile $t"
done
cd $root
python3 - $out/tiles/$wl "$@" > $out/tiles_$wl.txt <<'P'
import csv, glob, sys
from collections import defaultdict
base, tiles = sys.argv[1], sys.argv[2:]
print("plain (not residual-writing) launches of k_sweep_q, means per launch; share = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
for t in tiles:
    vals = defaultdict(list); dur = []; vg = None
    for fn in glob.glob(f"{base}/{t}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn, newline="")):
            k = row["Kernel_Name"]
            if "k_sweep_q<" in k and "k_sweep_q_redo" not in k and ", false, " in k.split("k_sweep_q<")[1][:40] and k.split("k_sweep_q<")[1].split(",")[2].strip() == "false":
                vals[row["Counter_Name"]].append(float(row["Counter_Value"])); vg = row["VGPR_Count"]
                if row["Counter_Name"] == "SQ_WAVE_CYCLES":
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    if not vals:
        print(t, "no data"); continue
    m = {k: sum(v) / len(v) for k, v in vals.items()}
    # (modal grid only: drop the short launch at the head of a block)
    print(f"tile {t}: vgpr {vg}  launches {len(dur)}  us/launch(under counters) {sorted(dur)[len(dur)//2]:.1f}  "
          f"WAIT_ANY/WAVE_CYCLES {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}  WAIT_INST_ANY/WAVE_CYCLES {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}  "
          f"VALU insts {m['SQ_INSTS_VALU']:.3e}  ACTIVE_INST_VALU {m['SQ_ACTIVE_INST_VALU']:.3e}  WAVE_CYCLES {m['SQ_WAVE_CYCLES']:.3e}  "
          f"BUSY_CYCLES {m['SQ_BUSY_CYCLES']:.3e}  LDS insts {m['SQ_INSTS_LDS']:.3e}  WAIT_INST_LDS {m['SQ_WAIT_INST_LDS']:.3e}")
P
cat $out/tiles_$wl.txt
