#!/bin/bash
# A/B sweep over prebuilt variants of liboctane_vof.so (octane_amd/variants/*.so): R1-style runs at 2000^2 and 5000^2 per variant.
mkdir -p gpurun_out/sweep
cp octane_amd/liboctane_vof.so /tmp/keep.so
for so in octane_amd/variants/*.so; do
  v=$(basename $so .so)
  cp $so octane_amd/liboctane_vof.so
  line="$v"
  for sz in "2000 6 20" "5000 8 10"; do
    set -- $sz
    python bench.py --size $1 --kiters $2 --steps $3 --warmup 3 --no-cpu-baseline --no-transfers > gpurun_out/sweep/ab.json 2>/dev/null
    line="$line $(python3 -c "import json; d=json.loads(open('gpurun_out/sweep/ab.json').read()); print($1, d['value'], d['roofline']['avg_launch_ms'])")"
  done
  echo "$line" | tee -a gpurun_out/sweep/results.txt
done
cp /tmp/keep.so octane_amd/liboctane_vof.so
