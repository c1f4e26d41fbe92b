#!/bin/bash
# round 6, GPU call 11: after the row rotation -- the whole GPU suite, the rocprofv3 trace + counter passes, every bench configuration, the soak
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rP -p no:cacheprovider > gpurun_out/r6_b11_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r6_b11_tests.txt
[ $rc -eq 0 ] || exit $rc
bash tools/profile_round.sh r6c
python tools/summarize_rocprof.py gpurun_out/prof_r6c gpurun_out/r6_kernel_trace_summary.md 5000 8 3 30 > gpurun_out/r6_summarize.log 2>&1; echo "summarize rc=$?"; tail -3 gpurun_out/r6_summarize.log
cp gpurun_out/prof_r6c/trace/*/*kernel_stats.csv gpurun_out/r6_kernel_stats.csv 2>/dev/null
mkdir -p gpurun_out/r6_pmc_raw
for c in fetch write; do f=$(ls gpurun_out/prof_r6c/$c/*/*counter_collection.csv | head -1); grep -E "Kernel_Name|octane" $f | gzip > gpurun_out/r6_pmc_raw/${c}_counter_collection_octane.csv.gz; done
rm -rf gpurun_out/prof_r6c/fetch gpurun_out/prof_r6c/write
bash tools/bench_all.sh r6c > gpurun_out/r6_bench_all.txt 2>&1; tail -9 gpurun_out/r6_bench_all.txt
timeout -k 10 300 python tools/soak_determinism.py 5000 20 > gpurun_out/r6_soak.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r6_soak.txt
