#!/bin/bash
# rocprofv3 kernel-trace stats of an arbitrary python tool.  Usage: tools/stats_cmd.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; rm -rf $out; mkdir -p $out
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $script "$@" > $out/stdout.txt 2> $out/stderr.txt
cd $root
tail -2 $out/stdout.txt
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f} {r['Percentage']:>6s}%")
P
