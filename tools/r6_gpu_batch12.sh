#!/bin/bash
# round 6, GPU call 12: counter + trace passes on the final sources (a declaration was added to vof_kernels.hpp, which the traffic guard hashes),
# then the default bench line with the traffic attached
mkdir -p gpurun_out
bash tools/profile_round.sh r6d
python tools/summarize_rocprof.py gpurun_out/prof_r6d gpurun_out/r6_kernel_trace_summary.md 5000 8 3 30 > gpurun_out/r6_summarize.log 2>&1; echo "summarize rc=$?"; tail -3 gpurun_out/r6_summarize.log
cp gpurun_out/prof_r6d/trace/*/*kernel_stats.csv gpurun_out/r6_kernel_stats.csv 2>/dev/null
mkdir -p gpurun_out/r6_pmc_raw
for c in fetch write; do f=$(ls gpurun_out/prof_r6d/$c/*/*counter_collection.csv | head -1); grep -E "Kernel_Name|octane" $f | gzip > gpurun_out/r6_pmc_raw/${c}_counter_collection_octane.csv.gz; done
rm -rf gpurun_out/prof_r6d/fetch gpurun_out/prof_r6d/write
cp gpurun_out/traffic.json profiles/traffic.json
timeout -k 10 400 python bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "drop-in", d["value_drop_in"], "roofline", r["achieved"], r["frac"], r["avg_launch_ms"], "traffic", r["traffic"], r.get("rocprof_traffic_frac"), "copy", r["frac_of_measured_copy"],
      "cpu", d["cpu_baseline"]["value"], {k: v["value"] for k, v in d["secondary"].items()})
PY
grep "k_pcg_fused .*7 (5000" gpurun_out/r6_kernel_trace_summary.md
