#!/bin/bash
# round 5, GPU call 8: band plans with a separate SHARED allocation (what neighbours read, the only part exported over IPC): row-band tests, the N = 2
# rehearsal of the multi-GPU side legs at full size
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_tiled_mp.py tests/test_gpu_persist.py "tests/test_gpu_fullsize.py::test_config3_quarter_scale_2712_four_bands_and_plain_match_oracle" \
   "tests/test_gpu_fullsize.py::test_config3_quarter_scale_disc_scene_four_bands_and_plain_match_oracle" tests/test_host_abi.py -m gpu -q -rP -p no:cacheprovider -x > gpurun_out/r5_b8_tests.txt 2>&1
echo "tests rc=$?"; grep "PARITY-FULLSIZE" gpurun_out/r5_b8_tests.txt | cut -c1-300; tail -3 gpurun_out/r5_b8_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
OCTANE_MP_TRACE=1 OCTANE_BENCH_SECONDARY_BUDGET_S=280 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r5_bench_rehearsal2_10848.json 2> gpurun_out/r5_bench_rehearsal2_10848.err
echo "rehearsal 10848 rc=$?"; cat gpurun_out/r5_bench_rehearsal2_10848.json | cut -c1-6000; grep "bench.py\|octane" gpurun_out/r5_bench_rehearsal2_10848.err | tail -24
timeout -k 10 200 python bench.py --workload tiled --bands 4 --steps 2 --warmup 1 > gpurun_out/r5_bench_tiled4.json 2> gpurun_out/r5_bench_tiled4.err
echo "tiled 4 virtual bands rc=$?"; cut -c1-400 gpurun_out/r5_bench_tiled4.json
timeout -k 10 100 python tools/ipc_probe.py 13.2 17 > gpurun_out/r5_ipc_probe2.txt 2>&1
echo "ipc probe rc=$?"; grep -v amdgpu.ids gpurun_out/r5_ipc_probe2.txt
