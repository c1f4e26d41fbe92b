#!/bin/bash
# round 6, GPU call 8: randomised parity sweeps on the final build (PRODUCT library), both scene families, fresh seeds
mkdir -p gpurun_out
timeout -k 10 700 python tools/fuzz_parity.py 150 61 > gpurun_out/r6_fuzz_lattice_150.txt 2>&1; echo "lattice rc=$?"; tail -2 gpurun_out/r6_fuzz_lattice_150.txt
OCTANE_FUZZ_FAMILY=disc timeout -k 10 400 python tools/fuzz_parity.py 80 62 > gpurun_out/r6_fuzz_disc_80.txt 2>&1; echo "disc rc=$?"; tail -2 gpurun_out/r6_fuzz_disc_80.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
