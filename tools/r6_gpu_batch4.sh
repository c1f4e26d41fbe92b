#!/bin/bash
# round 6, GPU call 4: the whole GPU suite on the reverse-walk build (+ the new recurrence test), then the bench line
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -rP -x -p no:cacheprovider --durations=25 > gpurun_out/r6_b4_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r6_b4_tests.txt; grep "RECURRENCE.*compared" gpurun_out/r6_b4_tests.txt
grep -A30 "slowest" gpurun_out/r6_b4_tests.txt | head -32
