#!/bin/bash
export TMPDIR=/tmp
run() { name=$1; shift; D=$PWD/gpurun_out/pmc2_$name; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- python bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline > $D/bench.log 2>&1
  echo "$name rc=$?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
python - <<'PY'
import csv,glob,collections
for name in ('sq','sq2','tcc','tcp'):
    fs=glob.glob(f'gpurun_out/pmc2_{name}/**/*_counter_collection.csv',recursive=True)
    if not fs: print(name,'no data'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'octane::' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('octane::')[1].split('(')[0]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in ('k_pcg_pass_a','k_pcg_pass_b','k_assemble'):
        for c,v in sorted(agg[k].items()):
            print(f'{name:4s} {k:14s} {c:30s} mean={sum(v)/len(v):.4g} n={len(v)}')
PY
