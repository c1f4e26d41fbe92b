#!/bin/bash
# round 5, GPU call 19: two more randomised parity sweeps on the final build (lattice family 200 cases, disc family 100 cases)
mkdir -p gpurun_out
timeout -k 10 560 python tools/fuzz_parity.py 200 113 > gpurun_out/r5_fuzz_lattice_200.txt 2>&1; echo "lattice rc=$?"; tail -1 gpurun_out/r5_fuzz_lattice_200.txt
OCTANE_FUZZ_FAMILY=disc timeout -k 10 560 python tools/fuzz_parity.py 100 7 > gpurun_out/r5_fuzz_disc_100.txt 2>&1; echo "disc rc=$?"; tail -1 gpurun_out/r5_fuzz_disc_100.txt
