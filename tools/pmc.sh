#!/bin/bash
# HBM traffic counters for the bench command, one counter set per pass (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Output under gpurun_out/pmc_<name>/.
export TMPDIR=/tmp
run() { name=$1; shift; D=$PWD/gpurun_out/pmc_$name; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $D/bench.log 2>&1
  echo "$name rc=$?"; ls $D/*/ | head -5; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES
