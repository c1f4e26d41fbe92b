#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the finest-level PCG kernel under a tuning environment: tools/pmc_fetch.sh TAG [VAR=value ...]
export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$PWD/gpurun_out/pmcf_$TAG; rm -rf $OUT; mkdir -p $OUT/f $OUT/w
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline --no-transfers > $OUT/f/bench.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline --no-transfers > $OUT/w/bench.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys
out, tag = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('f', 'w'):
    for f in glob.glob(f'{out}/{sub}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'octane::' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('octane::')[1].split('(')[0]
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    if 'fused_q' not in k: continue
    a = agg[k]
    m = lambda c: sum(a[c]) / len(a[c]) if a[c] else float('nan')
    # FETCH_SIZE / WRITE_SIZE are in kilobytes; FETCH_SIZE counts 128-byte requests as 64 on gfx950 (x2)
    rd, wr = m('FETCH_SIZE') * 1024 * 2, m('WRITE_SIZE') * 1024
    print(f"{tag:10s} {k:36s} read {rd / 1e6:8.1f} MB  write {wr / 1e6:7.1f} MB  n={len(a['FETCH_SIZE'])}")
PY
