#!/bin/bash
# round 6, GPU call 6: the prologue micro-benchmark, the rocprofv3 trace + counter passes of the final build, every bench
# configuration, the determinism soak
mkdir -p gpurun_out
timeout -k 10 120 tools/micro/bin/fold_probe > gpurun_out/r6_fold_probe.txt 2>&1; echo "fold_probe rc=$?"; cat gpurun_out/r6_fold_probe.txt
bash tools/profile_round.sh r6
python tools/summarize_rocprof.py gpurun_out/prof_r6 gpurun_out/r6_kernel_trace_summary.md 5000 8 3 30 > gpurun_out/r6_summarize.log 2>&1; echo "summarize rc=$?"; tail -3 gpurun_out/r6_summarize.log
cp gpurun_out/prof_r6/trace/*/*kernel_stats.csv gpurun_out/r6_kernel_stats.csv 2>/dev/null
mkdir -p gpurun_out/r6_pmc_raw
for c in fetch write; do f=$(ls gpurun_out/prof_r6/$c/*/*counter_collection.csv | head -1); grep -E "Kernel_Name|octane" $f | gzip > gpurun_out/r6_pmc_raw/${c}_counter_collection_octane.csv.gz; done
rm -rf gpurun_out/prof_r6/fetch gpurun_out/prof_r6/write
bash tools/bench_all.sh r6b > gpurun_out/r6_bench_all.txt 2>&1; tail -9 gpurun_out/r6_bench_all.txt
timeout -k 10 300 python tools/soak_determinism.py 5000 20 > gpurun_out/r6_soak.txt 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r6_soak.txt
du -sh gpurun_out/prof_r6 gpurun_out/r6_pmc_raw
