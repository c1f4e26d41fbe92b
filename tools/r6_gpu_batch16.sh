#!/bin/bash
# round 6, GPU call 16: end-of-round evidence on the final build -- a longer determinism soak and a mixed-family randomised sweep
mkdir -p gpurun_out
timeout -k 10 400 python tools/soak_determinism.py 5000 60 > gpurun_out/r6_soak60.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r6_soak60.txt
OCTANE_FUZZ_FAMILY=mixed timeout -k 10 900 python tools/fuzz_parity.py 200 81 > gpurun_out/r6_fuzz_mixed_200.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r6_fuzz_mixed_200.txt
