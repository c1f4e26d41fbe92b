#!/bin/bash
# round 6, GPU call 2: the reverse-walk variants on the levels that nearly fit the Infinity Cache (2500^2: level 6 of R1; 2000^2: configs[1])
mkdir -p gpurun_out
for n in 2500 2000; do
timeout -k 10 500 python tools/time_variants.py --size $n --reps 2 > gpurun_out/r6_rev_variants_$n.txt 2>&1
echo "variants $n rc=$?"; tail -9 gpurun_out/r6_rev_variants_$n.txt
done
