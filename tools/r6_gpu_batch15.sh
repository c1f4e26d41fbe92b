#!/bin/bash
# round 6, GPU call 15: the whole GPU suite + smoke() on the tree as committed at the end of the round
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -rP -p no:cacheprovider --durations=8 > gpurun_out/r6_b15_tests.txt 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r6_b15_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
