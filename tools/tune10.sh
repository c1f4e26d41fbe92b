#!/bin/bash
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for v in 0 1 0 1 0 1; do
  echo -n "GRAPH=$v: "
  OCTANE_TUNE_GRAPH=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'A',r['pass_a_ms'],'B',r['pass_b_ms'])"
done
