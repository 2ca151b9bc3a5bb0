#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py run.  Usage: tools/stats_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>7s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s}%")
P
cat $out/bench_under_rocprof.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])"
