mkdir -p gpurun_out
export OCTANE_LIB=$PWD/octane_amd/variants/asm_pack.so
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "assembly or lattice or channels or config0" 2>&1 | tail -3
for rep in 1 2; do for v in product asm_pack; do
  if [ $v = product ]; then unset OCTANE_LIB; else export OCTANE_LIB=$PWD/octane_amd/variants/$v.so; fi
  echo "== $v rep $rep: $(python tools/time_assembly.py 5000 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 2500 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 1250 2>&1 | grep 'FAST=1' | cut -c1-70)"
done; done > gpurun_out/r4_asm_pack.txt 2>&1; cat gpurun_out/r4_asm_pack.txt
for v in product asm_pack; do
  if [ $v = product ]; then unset OCTANE_LIB; else export OCTANE_LIB=$PWD/octane_amd/variants/$v.so; fi
  python bench.py --steps 10 --no-cpu-baseline --no-secondary --no-transfers > gpurun_out/r4_bench_$v.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r4_bench_$v.json').read().strip().splitlines()[-1]); print('$v R1', d['value'], d['ms_per_step'], d['roofline']['assemble_ms'], d['roofline']['setup_ms_all_levels'])"
done
