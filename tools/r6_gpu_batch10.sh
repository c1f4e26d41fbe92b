#!/bin/bash
# round 6, GPU call 10: tile columns rotated by the tile row (spreads the border-column tiles over all workgroups: at 5000^2 64 workgroups
# own ALL left-border tiles, 5 of their 24-25) against the shipped walk
mkdir -p gpurun_out
for n in 5000 2500; do
timeout -k 10 500 python tools/time_variants.py --size $n --reps 3 > gpurun_out/r6_rrot_$n.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r6_rrot_$n.txt
done
