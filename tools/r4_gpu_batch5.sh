mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_assembly_math.py tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | tail -4
for v in product asm_rows1 asm_rows2 asm_rows8; do
  if [ $v = product ]; then unset OCTANE_LIB; else export OCTANE_LIB=$PWD/octane_amd/variants/$v.so; fi
  echo "== $v"; python tools/time_assembly.py 5000 2>&1 | grep "FAST=1"; python tools/time_assembly.py 2500 2>&1 | grep "FAST=1"
done > gpurun_out/r4_asm_rows.txt 2>&1; cat gpurun_out/r4_asm_rows.txt
unset OCTANE_LIB
python bench.py --steps 10 --no-cpu-baseline --no-secondary --no-transfers > gpurun_out/r4_bench_asm_rows4.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r4_bench_asm_rows4.json').read().strip().splitlines()[-1]); print('R1', d['value'], d['ms_per_step'], d['roofline']['assemble_ms'])"
