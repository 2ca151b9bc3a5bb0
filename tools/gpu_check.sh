#!/bin/bash
# One GPU-box call: parity tests, the bench line, then the measurement pass.  Usage: tools/gpu_check.sh <tag> [workloads...]
set -e -o pipefail
tag=$1; shift
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/$tag/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/$tag/pytest_gpu.log
python bench.py --steps 2 --warmup 1 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -20 gpurun_out/$tag/bench.err; exit 1; }
cat gpurun_out/$tag/bench.json
bash tools/profile_round.sh $tag "$@"
