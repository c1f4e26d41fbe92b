#!/bin/bash
# kernel-trace of one bench step with and without the persistent mid-level solve: per-kernel, per-grid durations
export TMPDIR=/tmp
SIZE=${1:-5000}; KIT=${2:-8}
for mode in 1 0; do
  D=$PWD/gpurun_out/persist_trace_$mode; rm -rf $D; mkdir -p $D
  OCTANE_TUNE_PERSIST=$mode timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --size $SIZE --kiters $KIT --steps 2 --warmup 1 --no-cpu-baseline --no-transfers > $D/bench.log 2>&1
  echo "== OCTANE_TUNE_PERSIST=$mode: $(python3 -c "import json,sys; d=json.loads(open('$D/bench.log').read().strip().splitlines()[-1]); print(d['value'], 'Mpix/s', d['ms_per_step'], 'ms')" 2>/dev/null)"
  python3 tools/level_times.py $D solve_mid fused solve_small | awk '{ if ($6+0 < 200) print }'
done
