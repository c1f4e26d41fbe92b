#!/bin/bash
# round 5, GPU call 10: the whole GPU suite and smoke() on the final build
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b10_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r5_b10_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
