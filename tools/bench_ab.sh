run() { python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), d['unit'], round(d['roofline']['avg_launch_us'],1), 'us/launch')"; }
for c in 8x8 8x6 16x4 16x8 8x12; do echo -n "clinic q $c: "; XEE_TB_CFG=$c run --workload baroclinic; done
for t in 4 8; do echo -n "clinic q 8x6 T=$t: "; XEE_TB_T=$t XEE_TB_CFG=8x6 run --workload baroclinic; done
