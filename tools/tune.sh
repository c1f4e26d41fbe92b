#!/bin/bash
# developer sweep: single-level (full-resolution only) runs so the finest-level kernels dominate
for mb in 1024 2048; do for rb in 0 1; do
  echo "== MAXBLOCKS=$mb REVERSE_B=$rb"
  OCTANE_TUNE_MAXBLOCKS=$mb OCTANE_TUNE_REVERSE_B=$rb python bench.py --steps 2 --warmup 1 --kiters 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'])"
done; done
