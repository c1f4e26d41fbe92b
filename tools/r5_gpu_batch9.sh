#!/bin/bash
# round 5, GPU call 9: every bench configuration with the final build, the rocprofv3 trace + counter passes, the randomised parity sweep on the disc family
mkdir -p gpurun_out
bash tools/bench_all.sh > gpurun_out/r5_bench_all.txt 2>&1; echo "bench_all rc=$?"; tail -9 gpurun_out/r5_bench_all.txt
OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5b/b_rehearsal2.json 2> gpurun_out/r5b/b_rehearsal2.err; echo "rehearsal rc=$?"
bash tools/profile_round.sh r5
python tools/summarize_rocprof.py gpurun_out/prof_r5 gpurun_out/r5_kernel_trace_summary.md 5000 8 3 30 > gpurun_out/r5_summarize.log 2>&1; echo "summarize rc=$?"; tail -4 gpurun_out/r5_summarize.log
cp profiles/traffic.json gpurun_out/r5_traffic.json 2>/dev/null
python - <<'PY'
import csv, glob, gzip, os
out = "gpurun_out/r5_pmc_raw"; os.makedirs(out, exist_ok=True)
for sub in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/prof_r5/{sub}/**/*_counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if "octane::" in r.get("Kernel_Name", "")]
        if not keep: continue
        cols = [c for c in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value") if c in keep[0]]
        with gzip.open(f"{out}/{sub}_counter_collection.csv.gz", "wt", newline="") as g:
            w = csv.DictWriter(g, cols); w.writeheader()
            for r in keep:
                r = {c: r[c] for c in cols}
                r["Kernel_Name"] = r["Kernel_Name"].split("octane::")[1].split("(")[0][:60]
                w.writerow(r)
        print(sub, len(keep), "rows kept of", len(rows))
PY
OCTANE_FUZZ_FAMILY=disc timeout -k 10 600 python tools/fuzz_parity.py 60 53 > gpurun_out/r5_fuzz_disc.txt 2>&1; echo "fuzz disc rc=$?"; tail -2 gpurun_out/r5_fuzz_disc.txt
