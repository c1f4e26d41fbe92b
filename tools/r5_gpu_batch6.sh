#!/bin/bash
# round 5, GPU call 6: how long hipIpcOpenMemHandle takes on large allocations; the two-lane pair mode (batch tests, batch64 and --lanes 2 bench lines)
mkdir -p gpurun_out
timeout -k 10 240 python tools/ipc_probe.py 1 4 8 12 16 19 > gpurun_out/r5_ipc_probe.txt 2>&1
echo "ipc probe rc=$?"; cat gpurun_out/r5_ipc_probe.txt | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rP -p no:cacheprovider -k "batch" > gpurun_out/r5_b6_tests.txt 2>&1
echo "tests rc=$?"; grep "config4_batch64" gpurun_out/r5_b6_tests.txt | cut -c1-300; tail -2 gpurun_out/r5_b6_tests.txt
for rep in 1 2; do
timeout -k 10 300 python bench.py --workload batch64 --steps 3 --warmup 1 2>/dev/null | cut -c1-260
OCTANE_BENCH_LANE_MODE=0 timeout -k 10 300 python bench.py --workload batch64 --steps 3 --warmup 1 2>/dev/null | cut -c1-260
done > gpurun_out/r5_batch64.txt; cat gpurun_out/r5_batch64.txt
timeout -k 10 300 python bench.py --lanes 2 --steps 5 --warmup 2 2>/dev/null | cut -c1-260 > gpurun_out/r5_lanes2.txt; cat gpurun_out/r5_lanes2.txt
