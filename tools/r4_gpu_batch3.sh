set -o pipefail
mkdir -p gpurun_out
python bench.py --nchan 3 --steps 5 --warmup 2 > gpurun_out/r4_bench_nc3.json 2> gpurun_out/r4_bench_nc3.err; echo "nc3 rc=$?"; cut -c1-600 gpurun_out/r4_bench_nc3.json
python bench.py --nchan 2 --steps 5 --warmup 2 > gpurun_out/r4_bench_nc2.json 2> gpurun_out/r4_bench_nc2.err; echo "nc2 rc=$?"; cut -c1-300 gpurun_out/r4_bench_nc2.json
python tools/fuzz_parity.py 100 41 > gpurun_out/r4_fuzz_parity_100.txt 2>&1; echo "fuzz rc=$?"; tail -6 gpurun_out/r4_fuzz_parity_100.txt
python tools/runaway_check.py 10848 20240615 4 > gpurun_out/r4_config3_fullsize_vs_oracle.txt 2>&1; echo "config3 rc=$?"; grep -v amdgpu gpurun_out/r4_config3_fullsize_vs_oracle.txt | tail -12
