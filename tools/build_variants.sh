#!/bin/bash
# Builds the 32 variants of the LDS-DMA PCG kernel's text (octane_amd/csrc/pcg_fused_q_dma.hip: Q_P1 x Q_P2 x Q_LB x Q_ROT x
# scheduling strategy) as whole libraries under octane_amd/variants/ (git-ignored; they travel to the GPU box), each with its
# register count and occupancy printed.  tools/sweep_variants.sh then times them on the GPU.  Run `make -C octane_amd/csrc` first.
set -e
cd "$(dirname "$0")/../octane_amd/csrc"
rm -rf ../variants /tmp/octane_vb; mkdir -p ../variants /tmp/octane_vb
one() {
  n="v$1$2$3$4$5"; F=""; [ "$5" = "i" ] && F="-mllvm -amdgpu-sched-strategy=max-ilp"
  occ=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function $F \
        -DQ_P1=$1 -DQ_P2=$2 -DQ_LB=$3 -DQ_ROT=$4 -Rpass-analysis=kernel-resource-usage -c pcg_fused_q_dma.hip -o /tmp/octane_vb/$n.o 2>&1 |
        grep -E " VGPRs:|Occupancy|AGPRs:" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr -s ' ' | tr '\n' ' ')
  objs=$(ls *.o | grep -v pcg_fused_q_dma.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$n.so $objs /tmp/octane_vb/$n.o -lpthread
  echo "$n $occ"
}
export -f one
for p1 in 0 1; do for p2 in 0 1; do for lb in 1 2; do for rot in 0 1; do for sc in i n; do echo "$p1 $p2 $lb $rot $sc"; done; done; done; done; done |
  xargs -P 8 -L 1 bash -c 'one "$@"' _ | sort
