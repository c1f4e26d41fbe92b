#!/bin/bash
# round 6, GPU call 3: (a) more reverse-walk variants at 5000^2 (allocating tail of 5..24 rounds; streaming hint on the tail's dead p / r loads);
# (b) DRAM-side TCC counters next to the fabric-side ones on the R1 run, shipped text and reverse walk
mkdir -p gpurun_out
timeout -k 10 900 python tools/time_variants.py --size 5000 --reps 3 > gpurun_out/r6_rev_variants_b.txt 2>&1
echo "variants rc=$?"; tail -11 gpurun_out/r6_rev_variants_b.txt
export TMPDIR=/tmp
for lib in base rev1t8; do
export OCTANE_LIB=$PWD/octane_amd/variants/$lib.so
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_32B_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  D=$PWD/gpurun_out/r6_pmc_dram/$lib/$tag; mkdir -p $D
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $D -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-transfers --no-secondary > $D/bench.log 2>&1
  echo "$lib $tag rc=$?"
done
done
unset OCTANE_LIB
find gpurun_out/r6_pmc_dram -name "*.csv" | head -40
du -sh gpurun_out/r6_pmc_dram
