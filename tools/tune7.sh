#!/bin/bash
for v in 0 1 0 1 0 1; do
  echo -n "SMALL=$v: "
  OCTANE_TUNE_SMALL=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'Mpix/s',d['value'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'], 'finestPCG_ms', round(270*(r['pass_a_ms']+r['pass_b_ms']),1), 'rest_ms', round(d['ms_per_step']-270*(r['pass_a_ms']+r['pass_b_ms']),1))"
done
