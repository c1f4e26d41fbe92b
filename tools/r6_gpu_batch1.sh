#!/bin/bash
# round 6, GPU call 1: (a) which counters exist on the box (is there a DRAM-side TCC counter?), (b) the reverse-walk /
# allocating-tail variants of k_pcg_fused_q_dma against the shipped text
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(timeout -k 10 120 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r6_counters_avail.txt 2>&1; echo "list rc=$?")
cd $GRAFT_REPO_ROOT
grep -i -c "TCC" gpurun_out/r6_counters_avail.txt
grep -i -o "TCC_[A-Z0-9_]*DRAM[A-Z0-9_]*\|TCC_[A-Z0-9_]*MALL[A-Z0-9_]*\|TCC_EA0_[A-Z0-9_]*" gpurun_out/r6_counters_avail.txt | sort -u | tr '\n' ' '
echo
timeout -k 10 900 python tools/time_variants.py --size 5000 --reps 2 > gpurun_out/r6_rev_variants.txt 2>&1
echo "variants rc=$?"; tail -12 gpurun_out/r6_rev_variants.txt
