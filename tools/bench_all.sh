# Every bench.py configuration recorded under profiles/rN_bench_runs.jsonl, one GPU call (R3 stops some coarse solves early by the tolerance test: --allow-early-exit).
B=${1:-r6b}; mkdir -p gpurun_out/$B
python bench.py > gpurun_out/$B/b_r1.json 2> gpurun_out/$B/b_r1.err; tail -c 600 gpurun_out/$B/b_r1.json
python bench.py --size 2000 --kiters 6 --steps 20 --no-cpu-baseline > gpurun_out/$B/b_c1.json 2>/dev/null
python bench.py --liters 10 --cgiters 10 --steps 5 --no-cpu-baseline > gpurun_out/$B/b_r2.json 2>/dev/null
python bench.py --kiters 10 --liters 10 --steps 3 --warmup 1 --no-cpu-baseline --allow-early-exit > gpurun_out/$B/b_r3.json 2>/dev/null
python bench.py --workload batch64 --steps 2 --warmup 1 > gpurun_out/$B/b_c4.json 2>/dev/null
python bench.py --size 10848 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$B/b_10848.json 2>/dev/null
python bench.py --workload tiled --bands 4 --steps 2 --warmup 1 > gpurun_out/$B/b_c3.json 2> gpurun_out/$B/b_c3.err
python bench.py --lanes 2 --steps 5 --no-cpu-baseline > gpurun_out/$B/b_lanes2.json 2>/dev/null
for f in r1 c1 r2 r3 c4 10848 c3 lanes2; do python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/$B/b_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_launch_ms'))
except Exception as e: print('$f FAILED', e)
"; done
