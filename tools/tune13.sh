#!/bin/bash
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2 3 4 5; do for tr in 1 3; do
  echo -n "TRIALS=$tr: "
  OCTANE_TUNE_VERBOSE=1 OCTANE_TUNE_PLACEMENT_TRIALS=$tr python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['pass_a_ms'], r['pass_b_ms'])"
  grep octane /tmp/err.txt
done; done
