#!/bin/bash
# sustained shader clock under the sweep: GRBM_GUI_ACTIVE (cycles the GPU was busy) per dispatch over the dispatch's duration
set -e
root=$(pwd); out=$root/gpurun_out/clock; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 $root/tools/pmc_workload.py ${1:-ensemble} 800 > $out/log.txt 2>&1
cd $root
python3 - <<'P'
import csv, glob, statistics
f = glob.glob('gpurun_out/clock/**/*counter_collection.csv', recursive=True)[0]
v = []
for r in csv.DictReader(open(f)):
    if 'k_sweep_q<' in r['Kernel_Name'] and 'false, true, true' in r['Kernel_Name']:
        dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        v.append((float(r['Counter_Value']), dur))
v = v[10:]
print("dispatches", len(v), "median duration %.1f us" % statistics.median(d / 1e3 for _, d in v),
      "median GRBM_GUI_ACTIVE %.0f" % statistics.median(c for c, _ in v),
      "=> clock %.3f GHz" % statistics.median(c / d for c, d in v))
P
