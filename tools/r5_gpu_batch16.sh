#!/bin/bash
# round 5, GPU call 16: the default --gpus 2 run with its side legs as CHILD jobs (two ranks on the one GPU, gloo), at full size and with a leg made to crash
mkdir -p gpurun_out
OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5_bench_rehearsal2_10848.json 2> gpurun_out/r5_bench_rehearsal2_10848.err
echo "rehearsal 10848 rc=$?"; grep "bench.py: side leg" gpurun_out/r5_bench_rehearsal2_10848.err
# the same with the tiled leg's transport forced to something that does not exist: the child fails, the headline stands
OCTANE_TILED_TRANSPORT=nonsense OCTANE_BENCH_SECONDARY_TILED_SIZE=2712 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r5_bench_rehearsal2_failing_leg.json 2> gpurun_out/r5_bench_rehearsal2_failing_leg.err
echo "rehearsal with a failing leg rc=$?"; grep "bench.py: side leg" gpurun_out/r5_bench_rehearsal2_failing_leg.err
python - <<'PY'
import json
for f in ("gpurun_out/r5_bench_rehearsal2_10848.json", "gpurun_out/r5_bench_rehearsal2_failing_leg.json"):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln); s = d["secondary_multi_gpu"]
            print(f, d["value"], {k: (v.get("value"), v.get("ms_per_step"), v.get("exit_code"), v.get("leg_seconds"), (v.get("parity_vs_plain") or {}).get("rel_l2"), (v.get("transport") or {}).get("transport_used"), v.get("error")) for k, v in s.items()})
PY
