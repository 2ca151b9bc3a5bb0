#!/bin/bash
# Every bench workload once, on the GPU box: tools/bench_all.sh <tag>  ->  gpurun_out/<tag>/bench_<workload>.json (+ .err)
set -e
tag=$1
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench_ensemble.json 2> $out/bench_ensemble.err
for w in baroclinic single highres sweep sor; do
  timeout -k 10 400 python3 bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err
done
timeout -k 10 400 python3 bench.py --members 256 --steps 1 --warmup 0 --no-cpu-baseline > $out/bench_ensemble_256.json 2> $out/bench_ensemble_256.err
python3 - <<PY
import json
for w in ("ensemble", "baroclinic", "single", "highres", "sweep", "sor", "ensemble_256"):
    d = json.load(open("$out/bench_%s.json" % w)); r = d["roofline"]
    print(w, round(d["value"], 3), d["unit"], round(d["ms_per_step"], 2), "ms/step", r["bound"], round(r["frac"], 3), round(r.get("avg_launch_us") or 0, 1), "us",
          (r.get("traffic_source") or {}).get("matches_build"), (d.get("check") or {}).get("matches_golden"))
PY
