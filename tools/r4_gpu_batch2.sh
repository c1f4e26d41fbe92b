set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tiled_mp.py -x -q -m gpu --durations=12 > gpurun_out/r4_mp_tests.txt 2>&1; echo "mp tests rc=$?"; tail -18 gpurun_out/r4_mp_tests.txt | cut -c1-160
python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$?"; tail -2 gpurun_out/r4_bench_default.err
OCTANE_TUNE_MIN_BAND_PIXELS=1000000 python bench.py --workload tiled --bands 4 --size 2712 --steps 2 --warmup 1 > gpurun_out/r4_bench_tiled_2712.json 2> gpurun_out/r4_bench_tiled_2712.err; echo "tiled rc=$?"; grep -v amdgpu gpurun_out/r4_bench_tiled_2712.err | tail -3
for tr in auto collective; do
  if [ $tr = auto ]; then unset OCTANE_TILED_TRANSPORT; else export OCTANE_TILED_TRANSPORT=$tr; fi
  OCTANE_TUNE_MIN_BAND_PIXELS=1000000 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload tiled --size 2712 --steps 2 --warmup 1 > gpurun_out/r4_bench_tiled_mp_$tr.json 2> gpurun_out/r4_bench_tiled_mp_$tr.err; echo "tiled mp $tr rc=$?"; grep "bench.py" gpurun_out/r4_bench_tiled_mp_$tr.err | tail -3
done
unset OCTANE_TILED_TRANSPORT
python bench.py --workload tiled --bands 4 --steps 2 --warmup 1 > gpurun_out/r4_bench_tiled_10848.json 2> gpurun_out/r4_bench_tiled_10848.err; echo "tiled 10848 rc=$?"; grep "bench.py" gpurun_out/r4_bench_tiled_10848.err | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
