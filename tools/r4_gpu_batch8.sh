mkdir -p gpurun_out
export OCTANE_LIB=$PWD/octane_amd/variants/asm_waves8.so
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "assembly or lattice or channels or config0" 2>&1 | tail -2
for rep in 1 2; do for v in product asm_waves8 asm_waves16 asm_waves8_rows2; do
  if [ $v = product ]; then unset OCTANE_LIB; else export OCTANE_LIB=$PWD/octane_amd/variants/$v.so; fi
  echo "== $v rep $rep: $(python tools/time_assembly.py 5000 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 2500 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 1250 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 625 2>&1 | grep 'FAST=1' | cut -c1-70)"
done; done > gpurun_out/r4_asm_waves.txt 2>&1; cat gpurun_out/r4_asm_waves.txt
