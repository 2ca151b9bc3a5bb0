#!/bin/bash
# PMC passes only (no bench stats).  Usage: tools/pmc_only.sh <tag> [workloads...]
set -e
tag=$1; shift
wls=${@:-ensemble}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
python3 $root/tools/build_hash.py > $out/build_hash.txt
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"
SQ2="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES"
for wl in $wls; do
  d=$out/pmc/$wl; mkdir -p $d
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $root/tools/pmc_workload.py $wl > $d/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 $root/tools/pmc_workload.py $wl > $d/write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $d/sq1 -- python3 $root/tools/pmc_workload.py $wl > $d/sq1.log 2>&1 || echo "sq1 failed"
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $d/sq2 -- python3 $root/tools/pmc_workload.py $wl > $d/sq2.log 2>&1 || echo "sq2 failed"
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $d/tcc -- python3 $root/tools/pmc_workload.py $wl > $d/tcc.log 2>&1 || echo "tcc failed"
done
cd $root
python3 tools/pmc_summarize.py $out $out/summary > $out/summary.log 2>&1 || echo "summarize failed"
python3 - $out <<'P'
import json, sys
d = json.load(open(sys.argv[1] + "/summary/pmc_summary.json"))
for wl, r in d.items():
    keys = ["dispatches", "vgpr", "FETCH_SIZE_kib", "WRITE_SIZE_kib", "bytes_per_launch", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"]
    print(wl, r["kernel"][:60])
    print("   ", {k: (round(r[k], 1) if isinstance(r.get(k), float) else r.get(k)) for k in keys})
P
