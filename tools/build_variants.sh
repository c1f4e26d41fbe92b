#!/bin/bash
# Builds named variants of the LDS-DMA PCG kernel (octane_amd/csrc/pcg_fused_q_dma.hip) as whole libraries under octane_amd/variants/
# (git-ignored; they travel to the GPU box), each with its register counts.  tools/time_variants.py times them on the GPU
# (OCTANE_LIB=<variant>.so selects the library).  Run `make -C octane_amd/csrc DIAG=1` first.
# usage: tools/build_variants.sh "name:flags" ...   e.g.  "nop2:-DQ_P2=0" "norot:-DQ_ROT=0 -mllvm -amdgpu-sched-strategy=max-ilp"
# (round 5: the default-off EXPERIMENT switches of rounds 2-4 -- Q_ABL ablations, Q_LASTFOLD, Q_TOUCH -- were removed from the kernel, the object
# byte-identical before and after; their measurements are in EXPERIMENTS.md 5 / 8 and profiles/r3_ablation_q_dma.txt, r4_lastfold.txt.  The text
# switches that DEFINE the shipped kernel -- Q_P1, Q_P2, Q_LB, Q_A2BR, Q_VMN, Q_ROT -- remain.)
set -e
cd "$(dirname "$0")/../octane_amd/csrc"
mkdir -p ../variants /tmp/octane_vb
one() {
  n="${1%%:*}"; f="${1#*:}"
  # variants are DIAGNOSTIC-flavoured libraries (round 6): octane_vof_plan_probe / octane_vof_tune, which tools/time_variants.py needs, are not
  # exported by the product library any more -- so: the diagnostic build's objects (make DIAG=1) + this variant of the kernel with -DOCTANE_DIAG=1
  occ=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function -DOCTANE_DIAG=1 $f \
        -Rpass-analysis=kernel-resource-usage -c pcg_fused_q_dma.hip -o /tmp/octane_vb/$n.o 2>&1 |
        grep -E " VGPRs:|Occupancy|AGPRs:" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr -s ' ' | tr '\n' ' ')
  objs=$(ls *.diag.o | grep -v "pcg_fused_q_dma.diag.o" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../variants/$n.so $objs /tmp/octane_vb/$n.o -lpthread
  echo "$n [$f] $occ"
}
export -f one
printf '%s\n' "$@" | xargs -P 6 -I{} bash -c 'one "$@"' _ {} | sort
