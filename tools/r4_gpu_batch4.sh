set -o pipefail
mkdir -p gpurun_out
python bench.py --nchan 3 --steps 5 --warmup 2 --allow-early-exit > gpurun_out/r4_bench_nc3.json 2> gpurun_out/r4_bench_nc3.err; echo "nc3 rc=$?"; cut -c1-700 gpurun_out/r4_bench_nc3.json
python bench.py --nchan 2 --steps 5 --warmup 2 --allow-early-exit > gpurun_out/r4_bench_nc2.json 2> gpurun_out/r4_bench_nc2.err; echo "nc2 rc=$?"; cut -c1-300 gpurun_out/r4_bench_nc2.json
OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r4_bench_pair_2ranks_rehearsal.json 2> gpurun_out/r4_bench_pair_2ranks_rehearsal.err; echo "pair rehearsal rc=$?"; cut -c1-400 gpurun_out/r4_bench_pair_2ranks_rehearsal.json
python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite_final.txt 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r4_gpu_suite_final.txt
