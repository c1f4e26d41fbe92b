#!/bin/bash
# round 5, GPU call 3: solo-band timing, the default bench line, the N = 2 rehearsal of the multi-GPU side legs (two ranks on the one GPU, gloo), the whole GPU suite
mkdir -p gpurun_out
export OCTANE_LIB=$PWD/octane_amd/liboctane_vof_diag.so
timeout -k 10 600 python tools/solo_band.py 10848 8 3 30 2,4,8 > gpurun_out/r5_solo_band_10848.txt 2>&1
echo "solo 10848 rc=$?"; cat gpurun_out/r5_solo_band_10848.txt
timeout -k 10 300 python tools/solo_band.py 5000 8 3 30 2,4,8 > gpurun_out/r5_solo_band_5000.txt 2>&1
echo "solo 5000 rc=$?"; cat gpurun_out/r5_solo_band_5000.txt
unset OCTANE_LIB
timeout -k 10 400 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r5_bench_default.json; tail -3 gpurun_out/r5_bench_default.err
OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5_bench_rehearsal2.json 2> gpurun_out/r5_bench_rehearsal2.err
echo "rehearsal rc=$?"; cut -c1-3000 gpurun_out/r5_bench_rehearsal2.json; tail -8 gpurun_out/r5_bench_rehearsal2.err
python -m pytest tests -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b3_tests.txt 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5_b3_tests.txt
