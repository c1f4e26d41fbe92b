#!/bin/bash
# round 5, GPU call 15: event-ordered phase boundaries in the process form of the row bands: the multi-process tests, the N = 2 rehearsals at 2712^2 and 10848^2
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tiled_mp.py -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b15_tests.txt 2>&1; echo "mp tests rc=$?"; tail -3 gpurun_out/r5_b15_tests.txt
OCTANE_BENCH_SECONDARY_TILED_SIZE=2712 OCTANE_TUNE_MIN_BAND_PIXELS=1000000 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5_bench_rehearsal2_2712.json 2> gpurun_out/r5_bench_rehearsal2_2712.err
echo "rehearsal 2712 rc=$?"; grep "bench.py tiled" gpurun_out/r5_bench_rehearsal2_2712.err | tail -3
OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5_bench_rehearsal2_10848.json 2> gpurun_out/r5_bench_rehearsal2_10848.err
echo "rehearsal 10848 rc=$?"; grep "bench.py tiled\|octane:" gpurun_out/r5_bench_rehearsal2_10848.err | tail -5
python - <<'PY'
import json
for f in ("gpurun_out/r5_bench_rehearsal2_2712.json", "gpurun_out/r5_bench_rehearsal2_10848.json"):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln); s = d["secondary_multi_gpu"]
            print(f, d["value"], {k: (v.get("value"), v.get("ms_per_step"), (v.get("parity_vs_plain") or {}).get("rel_l2"), (v.get("transport") or {}).get("transport_used")) for k, v in s.items() if isinstance(v, dict)})
PY
