#!/bin/bash
# round 5, GPU call 4: the reference's own kernel text (hipified, oracle/_ref) as a cross-check; solo-band timing with mirrored halos; the N = 2 rehearsal of
# the multi-GPU side legs at 2712^2 and at full size; the file-layer tests; rocprofv3 trace + counter passes of the bench command
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_refhip.py tests/test_io_nc4.py tests/test_host_abi.py -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b4_tests.txt 2>&1
echo "tests rc=$?"; grep "REFHIP" gpurun_out/r5_b4_tests.txt | cut -c1-400; tail -3 gpurun_out/r5_b4_tests.txt
export OCTANE_LIB=$PWD/octane_amd/liboctane_vof_diag.so
timeout -k 10 600 python tools/solo_band.py 10848 8 3 30 2,4,8 > gpurun_out/r5_solo_band_10848.txt 2>&1
echo "solo 10848 rc=$?"; cat gpurun_out/r5_solo_band_10848.txt
timeout -k 10 300 python tools/solo_band.py 5000 8 3 30 2,4,8 > gpurun_out/r5_solo_band_5000.txt 2>&1
echo "solo 5000 rc=$?"; cat gpurun_out/r5_solo_band_5000.txt
unset OCTANE_LIB
OCTANE_BENCH_SECONDARY_TILED_SIZE=2712 OCTANE_TUNE_MIN_BAND_PIXELS=1000000 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5_bench_rehearsal2_2712.json 2> gpurun_out/r5_bench_rehearsal2_2712.err
echo "rehearsal 2712 rc=$?"; cut -c1-200 gpurun_out/r5_bench_rehearsal2_2712.json; grep "bench.py" gpurun_out/r5_bench_rehearsal2_2712.err | tail -12
OCTANE_BENCH_SECONDARY_BUDGET_S=240 OCTANE_BENCH_ONE_DEVICE=1 OCTANE_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r5_bench_rehearsal2_10848.json 2> gpurun_out/r5_bench_rehearsal2_10848.err
echo "rehearsal 10848 rc=$?"; cut -c1-200 gpurun_out/r5_bench_rehearsal2_10848.json; grep "bench.py\|octane:" gpurun_out/r5_bench_rehearsal2_10848.err | tail -12
bash tools/profile_round.sh r5
python tools/summarize_rocprof.py gpurun_out/prof_r5 gpurun_out/r5_kernel_trace_summary.md 5000 8 3 30 > gpurun_out/r5_summarize.log 2>&1; echo "summarize rc=$?"; tail -3 gpurun_out/r5_summarize.log
# keep the raw counter passes (octane kernels only) small enough to travel back and be committed
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
ls -la gpurun_out/r5_pmc_raw gpurun_out/prof_r5/trace | head; du -sh gpurun_out/prof_r5
