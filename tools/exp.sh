#!/bin/bash
cp octane_amd/liboctane_vof.so /tmp/keep.so
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2; do for v in base nodiv nohalo both; do
  cp octane_amd/liboctane_exp_$v.so octane_amd/liboctane_vof.so
  echo -n "$v: "
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'A',r['pass_a_ms'],'B',r['pass_b_ms'])"
done; done
cp /tmp/keep.so octane_amd/liboctane_vof.so
