#!/bin/bash
# Builds named variants of the LDS-DMA PCG kernel (octane_amd/csrc/pcg_fused_q_dma.hip) as whole libraries under octane_amd/variants/
# (git-ignored; they travel to the GPU box), each with its register counts.  tools/time_variants.py times them on the GPU
# (OCTANE_LIB=<variant>.so selects the library).  Run `make -C octane_amd/csrc` first.
# usage: tools/build_variants.sh "name:flags" ...   e.g.  "t2n:-DQ_TOUCH=1" "abl1:-DQ_ABL=1 -mllvm -amdgpu-sched-strategy=max-ilp"
set -e
cd "$(dirname "$0")/../octane_amd/csrc"
mkdir -p ../variants /tmp/octane_vb
one() {
  n="${1%%:*}"; f="${1#*:}"
  occ=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function $f \
        -Rpass-analysis=kernel-resource-usage -c pcg_fused_q_dma.hip -o /tmp/octane_vb/$n.o 2>&1 |
        grep -E " VGPRs:|Occupancy|AGPRs:" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr -s ' ' | tr '\n' ' ')
  objs=$(ls *.o | grep -v "pcg_fused_q_dma.o\|\.diag\.o" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$n.so $objs /tmp/octane_vb/$n.o -lpthread
  echo "$n [$f] $occ"
}
export -f one
printf '%s\n' "$@" | xargs -P 6 -I{} bash -c 'one "$@"' _ {} | sort
