mkdir -p gpurun_out
for rep in 1 2; do for v in asm_rows1 asm_rows4 asm_rows6 asm_rows8 asm_rows12 asm_rows16; do
  if [ $v = asm_rows4 ]; then unset OCTANE_LIB; else export OCTANE_LIB=$PWD/octane_amd/variants/$v.so; fi
  echo "== $v rep $rep: $(python tools/time_assembly.py 5000 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 2500 2>&1 | grep 'FAST=1' | cut -c1-70) | $(python tools/time_assembly.py 1250 2>&1 | grep 'FAST=1' | cut -c1-70)"
done; done > gpurun_out/r4_asm_rows2.txt 2>&1; cat gpurun_out/r4_asm_rows2.txt
