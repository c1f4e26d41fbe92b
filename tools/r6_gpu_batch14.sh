#!/bin/bash
# round 6, GPU call 14: the new tile-walk sweep test, then fresh randomised parity sweeps on the FINAL build (row rotation included)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest "tests/test_gpu_parity.py::test_tile_walks_of_the_lds_dma_kernel_on_random_large_shapes" -m gpu -q -rP -p no:cacheprovider > gpurun_out/r6_b14_walks.txt 2>&1; echo "walks rc=$?"; grep "PARITY case=tile_walks" gpurun_out/r6_b14_walks.txt | cut -c1-230; tail -1 gpurun_out/r6_b14_walks.txt
timeout -k 10 500 python tools/fuzz_parity.py 100 71 > gpurun_out/r6_fuzz_lattice_100b.txt 2>&1; echo "lattice rc=$?"; tail -1 gpurun_out/r6_fuzz_lattice_100b.txt
OCTANE_FUZZ_FAMILY=disc timeout -k 10 400 python tools/fuzz_parity.py 60 72 > gpurun_out/r6_fuzz_disc_60b.txt 2>&1; echo "disc rc=$?"; tail -1 gpurun_out/r6_fuzz_disc_60b.txt
