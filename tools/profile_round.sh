#!/bin/bash
# Measurement pass on the GPU box.  Usage: tools/profile_round.sh <tag> [workloads...]
#   1. kernel-trace stats of the default bench and of --workload baroclinic (rocprofv3 --kernel-trace --stats)
#   2. per workload: FETCH_SIZE pass, WRITE_SIZE pass (separate: TCC has 4 slots, the two need 3 + 2) and two SQ
#      passes on the PRODUCTION kernels (tools/pmc_workload.py) -- counters only, no trace domains besides
#      --kernel-trace (gpurun refuses --pmc mixed with sys/hip/hsa traces)
# Everything lands in gpurun_out/<tag>/; tools/pmc_summarize.py condenses it into profiles/<round>/.
set -e
tag=$1; shift
wls=${@:-ensemble baroclinic single highres}
root=$(pwd)
out=$root/gpurun_out/$tag
rm -rf $out/stats $out/stats_baroclinic $out/pmc $out/summary
mkdir -p $out
python3 $root/tools/build_hash.py > $out/build_hash.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_baroclinic -- python3 $root/bench.py --workload baroclinic --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_baroclinic_under_rocprof.json 2> $out/bench_baroclinic_under_rocprof.err
echo "stats done"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"
SQ2="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES"
for wl in $wls; do
  d=$out/pmc/$wl
  mkdir -p $d
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $root/tools/pmc_workload.py $wl > $d/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 $root/tools/pmc_workload.py $wl > $d/write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $d/sq1 -- python3 $root/tools/pmc_workload.py $wl > $d/sq1.log 2>&1 || echo "sq1 pass failed for $wl (see $d/sq1.log)"
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $d/sq2 -- python3 $root/tools/pmc_workload.py $wl > $d/sq2.log 2>&1 || echo "sq2 pass failed for $wl (see $d/sq2.log)"
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $d/tcc -- python3 $root/tools/pmc_workload.py $wl > $d/tcc.log 2>&1 || echo "tcc pass failed for $wl (see $d/tcc.log)"
  echo "pmc $wl done"
done
cd $root
# the raw per-dispatch CSVs are large: keep only what the summary needs (the sweep + calibration kernels)
python3 tools/pmc_summarize.py $out $out/summary > $out/summary.log 2>&1 || echo "summarize failed"
find $out -name "*counter_collection.csv" -size +4M -delete
find $out -name "*kernel_trace.csv" -size +4M -delete
find $out -name "*kernel_stats.csv" | head -3
