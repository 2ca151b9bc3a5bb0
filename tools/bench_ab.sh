run() { python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), d['unit'], round(d['ms_per_step'],1), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')"; }
python -m pytest tests/test_gpu_quotient.py tests/test_gpu_parity.py tests/test_gpu_pipelined_checks.py -x -q 2>&1 | tail -1
echo -n "ens: "; run
echo -n "single: "; run --workload single
echo -n "clinic: "; run --workload baroclinic
