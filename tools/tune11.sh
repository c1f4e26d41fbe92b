#!/bin/bash
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for cfg in "0 2048" "1 2048" "0 1024" "0 2048" "1 2048" "0 1024"; do set -- $cfg
  echo -n "XCD=$1 MAXBLOCKS=$2: "
  OCTANE_TUNE_XCD=$1 OCTANE_TUNE_MAXBLOCKS=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'Mpix/s',d['value'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'])"
done
