#!/bin/bash
export TMPDIR=/tmp
for v in 0 2 0 2; do
  echo -n "PASS_A=$v: "
  OCTANE_TUNE_PASS_A=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('ms/step',d['ms_per_step'],'Mpix/s',d['value'],'A',r['pass_a_ms'],'B',r['pass_b_ms'],'asm',r['assemble_ms'])"
done
for v in 2; do
  D=$PWD/gpurun_out/pmc_pa$v; rm -rf $D; mkdir -p $D
  OCTANE_TUNE_PASS_A=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -- python bench.py --steps 1 --warmup 0 --kiters 1 --no-cpu-baseline > $D/bench.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('$D/**/*_counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'octane::' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        agg[r['Kernel_Name'].split('octane::')[1].split('(')[0]].append(float(r['Counter_Value']))
for k,v in agg.items():
    if k.startswith('k_pcg'): print('PASS_A=$v', k, 'mean FETCH_SIZE x2 MB: %.0f' % (sum(v)/len(v)*1024*2/1e6))
PY
done
