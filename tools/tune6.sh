#!/bin/bash
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for v in 0 4 5 0 4 5; do
  echo -n "PASS_A=$v: "
  OCTANE_TUNE_PASS_A=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'Mpix/s',d['value'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'])"
done
