#!/bin/bash
# Occupancy / stall / instruction-mix counters of the finest-level PCG kernel (k_pcg_fused_q at 5000^2), one rocprofv3 --pmc
# pass per counter group (counters only: no sys / hip / hsa tracing next to --pmc on this pool).  Output: a table on stdout.
# usage: tools/pmc_r2.sh TAG [extra bench.py args]
export TMPDIR=/tmp
TAG=${1:-r2}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
run() { name=$1; shift; D=$OUT/$name; rm -rf $D; mkdir -p $D
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- python3 bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline --no-transfers $EXTRA > $D/bench.log 2>&1
  echo "# $name rc=$?"; }
EXTRA="$@"
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for name in ('sq', 'sq2', 'sq3', 'tcc'):
    fs = glob.glob(f'{out}/{name}/**/*_counter_collection.csv', recursive=True)
    if not fs:
        print(name, 'no data'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'octane::' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('octane::')[1].split('(')[0]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(agg):
        if not any(t in k for t in ('fused', 'assemble')):
            continue
        for c, v in sorted(agg[k].items()):
            print(f'{name:4s} {k:44s} {c:28s} mean={sum(v) / len(v):.5g} n={len(v)}')
PY
