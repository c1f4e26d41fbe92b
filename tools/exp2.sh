#!/bin/bash
cp octane_amd/liboctane_vof.so /tmp/keep.so
for rep in 1 2; do for v in base noside norows nohalo nowys nodiv all; do
  cp octane_amd/liboctane_exp_$v.so octane_amd/liboctane_vof.so
  echo "== $v"
  OCTANE_TUNE_PLACEMENT_TRIALS=1 python tools/probe_levels.py 2>/dev/null | tail -3
done; done
cp /tmp/keep.so octane_amd/liboctane_vof.so
