#!/bin/bash
# end-of-round measurement pass on the GPU box: kernel-trace stats of the default bench, then the two PMC passes
# (separate runs, no trace domains mixed with --pmc).  Usage: tools/profile_round.sh <tag>
set -e
tag=$1
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/tools/pmc_case.py 1024 512 32 > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/tools/pmc_case.py 1024 512 32 > $out/pmc_write.log 2>&1
cd $root
find $out -name "*kernel_stats.csv" | head -3
