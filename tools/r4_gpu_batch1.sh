set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pix2uv.py tests/test_gpu_persist.py tests/test_gpu_assembly_math.py tests/test_gpu_parity.py -x -q -rP -m gpu > gpurun_out/r4_tests_b1.txt 2>&1; echo "tests rc=$?"; grep -E "PIX2UV-FMAD|passed|failed|Error" gpurun_out/r4_tests_b1.txt | tail -5
python tools/time_assembly.py --nc 5000 > gpurun_out/r4_time_assembly_nc.txt 2>&1; cat gpurun_out/r4_time_assembly_nc.txt
python tools/time_assembly.py 5000 > gpurun_out/r4_time_assembly.txt 2>&1; cat gpurun_out/r4_time_assembly.txt
./tools/micro/bin/h2d_pitch 5000 > gpurun_out/r4_h2d_pitch.txt 2>&1; cat gpurun_out/r4_h2d_pitch.txt
python bench.py --workload tiled --bands 4 --size 2712 --steps 2 --warmup 1 > gpurun_out/r4_bench_tiled_2712.json 2> gpurun_out/r4_bench_tiled_2712.err; echo "tiled rc=$?"; tail -2 gpurun_out/r4_bench_tiled_2712.err; cut -c1-1500 gpurun_out/r4_bench_tiled_2712.json
OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload tiled --size 2712 --steps 2 --warmup 1 > gpurun_out/r4_bench_tiled_mp.json 2> gpurun_out/r4_bench_tiled_mp.err; echo "tiled mp rc=$?"; grep -v amdgpu.ids gpurun_out/r4_bench_tiled_mp.err | tail -4; cut -c1-1800 gpurun_out/r4_bench_tiled_mp.json
for mbp in 12582912 6000000 1500000 400000; do echo "== min_band_pixels $mbp"; python tools/tiled_virtual.py 10848 8 3 30 4 $mbp 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4_tiled_threshold.txt; cat gpurun_out/r4_tiled_threshold.txt
