#!/bin/bash
# A/B two builds of the library in one GPU session: octane_amd/liboctane_vof_old.so vs liboctane_vof.so
cp octane_amd/liboctane_vof.so /tmp/new.so
for rep in 1 2; do for which in old new; do
  if [ $which = old ]; then cp octane_amd/liboctane_vof_old.so octane_amd/liboctane_vof.so; else cp /tmp/new.so octane_amd/liboctane_vof.so; fi
  echo -n "$which: "
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'Mpix/s',d['value'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'])"
done; done
cp /tmp/new.so octane_amd/liboctane_vof.so
