run() { python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), d['unit'], round(d['roofline']['avg_launch_us'],1), 'us/launch')"; }
for t in 4 5 7 8 10; do echo -n "slim cap6 rows=6 T=$t: "; XEE_TB_T=$t XEE_S_ROWS=6 run; done
for t in 5 7 10; do echo -n "q 8x8 T=$t: "; XEE_TB_T=$t XEE_NO_SLIM=1 run; done
