#!/bin/bash
# Produces the rocprofv3 artefacts kept under profiles/: kernel-trace + stats of the bench command, and the
# HBM traffic counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
export TMPDIR=/tmp
TAG=${1:-r1}
D=$PWD/gpurun_out/prof_$TAG; rm -rf $D; mkdir -p $D/trace $D/fetch $D/write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-transfers --no-secondary > $D/trace/bench.log 2>&1; echo "trace rc=$?"; tail -1 $D/trace/bench.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-transfers --no-secondary > $D/fetch/bench.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-transfers --no-secondary > $D/write/bench.log 2>&1; echo "write rc=$?"
