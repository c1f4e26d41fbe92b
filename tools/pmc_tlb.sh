#!/bin/bash
# Address-translation and fabric-latency counters of the finest-level PCG kernel, with its duration from the same run: boxes of
# this pool run the same binary at 0.33 or 0.42 ms per launch, and this is to tell what differs.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_tlb; rm -rf $OUT; mkdir -p $OUT/a $OUT/b
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --output-format csv -d $OUT/a -- python3 bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline --no-transfers > $OUT/a/bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum --output-format csv -d $OUT/b -- python3 bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline --no-transfers > $OUT/b/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
for sub in ('a', 'b'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{sub}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_pcg_fused_q' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('octane::')[1].split('(')[0]
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(f'{out}/{sub}/**/*_kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_pcg_fused_q' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('octane::')[1].split('(')[0]
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k in sorted(agg):
        print(sub, k, f"mean duration {sum(dur[k]) / len(dur[k]):.1f} us (under the counters)", {c: f"{sum(v) / len(v):.4g}" for c, v in sorted(agg[k].items())})
    try:
        d = json.loads(open(f'{out}/{sub}/bench.log').read().strip().splitlines()[-1])
        print(sub, 'placement trials', d.get('placement_trials_ms'), 'avg launch', d['roofline']['avg_launch_ms'])
    except Exception as e:
        print('no bench line', e)
PY
