#!/bin/bash
for k in 1 8 1 8; do for cfg in "1024 0" "2048 1"; do set -- $cfg
  echo -n "kiters=$k MAXBLOCKS=$1 REVERSE_B=$2: "
  OCTANE_TUNE_MAXBLOCKS=$1 OCTANE_TUNE_REVERSE_B=$2 python bench.py --steps 2 --warmup 1 --kiters $k --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'])"
done; done
