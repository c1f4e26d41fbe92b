#!/bin/bash
# round 6, GPU call 7: what a sums-only sweep costs (odd launches without their r / p stores): the first number behind "two iterations per sweep"
mkdir -p gpurun_out
timeout -k 10 600 python tools/time_variants.py --size 5000 --reps 3 > gpurun_out/r6_nost.txt 2>&1; echo rc=$?; tail -5 gpurun_out/r6_nost.txt
OCTANE_LIB=$PWD/octane_amd/variants/base.so timeout -k 10 200 python tools/launch_kinds.py > gpurun_out/r6_kinds_base.txt 2>&1; tail -1 gpurun_out/r6_kinds_base.txt
