#!/bin/bash
# round 5, GPU call 5: hipified-reference cross-check (one HIP runtime per process now), solo-band timing with the stop test held open,
# where the 10848^2 process-form creation spends its time, the lanes experiment
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_refhip.py tests/test_io_nc4.py tests/test_host_abi.py -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b5_tests.txt 2>&1
echo "tests rc=$?"; grep "REFHIP" gpurun_out/r5_b5_tests.txt | cut -c1-500; tail -3 gpurun_out/r5_b5_tests.txt
export OCTANE_LIB=$PWD/octane_amd/liboctane_vof_diag.so
timeout -k 10 600 python tools/solo_band.py 10848 8 3 30 2,4,8 > gpurun_out/r5_solo_band_10848.txt 2>&1
echo "solo 10848 rc=$?"; cat gpurun_out/r5_solo_band_10848.txt
timeout -k 10 300 python tools/solo_band.py 5000 8 3 30 2,4,8 > gpurun_out/r5_solo_band_5000.txt 2>&1
echo "solo 5000 rc=$?"; cat gpurun_out/r5_solo_band_5000.txt
unset OCTANE_LIB
timeout -k 10 300 python tools/lanes_concurrent.py 2000 6 8 > gpurun_out/r5_lanes_concurrent.txt 2>&1
echo "lanes rc=$?"; cat gpurun_out/r5_lanes_concurrent.txt
OCTANE_MP_TRACE=1 OCTANE_BENCH_SECONDARY_BUDGET_S=200 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r5_bench_rehearsal2_10848.json 2> gpurun_out/r5_bench_rehearsal2_10848.err
echo "rehearsal 10848 rc=$?"; cut -c1-200 gpurun_out/r5_bench_rehearsal2_10848.json; grep "bench.py\|octane" gpurun_out/r5_bench_rehearsal2_10848.err | tail -20
