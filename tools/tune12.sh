#!/bin/bash
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2 3 4; do for cfg in "0 0" "2097152 0" "2097152 64" "65536 0" "4096 0"; do set -- $cfg
  echo -n "ALIGN=$1 SKEW=$2: "
  OCTANE_TUNE_PLANE_ALIGN=$1 OCTANE_TUNE_SKEW=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['pass_a_ms'], r['pass_b_ms'])"
done; done
