#!/bin/bash
# round 5, GPU call 12: more randomised parity (both scene families), the determinism soak on the final build
mkdir -p gpurun_out
OCTANE_FUZZ_FAMILY=mixed timeout -k 10 800 python tools/fuzz_parity.py 120 71 > gpurun_out/r5_fuzz_mixed.txt 2>&1; echo "fuzz mixed rc=$?"; tail -2 gpurun_out/r5_fuzz_mixed.txt
timeout -k 10 300 python tools/soak_determinism.py 5000 40 > gpurun_out/r5_soak.txt 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r5_soak.txt
