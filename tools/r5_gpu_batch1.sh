#!/bin/bash
# round 5, GPU call 1: data-shaped parity (tests + the tool's table with the growth curve)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_disc.py -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_disc_tests.txt 2>&1
echo "disc tests rc=$?" 
tail -5 gpurun_out/r5_disc_tests.txt
timeout -k 10 900 python tools/disc_parity.py full growth > gpurun_out/r5_disc_parity.txt 2>&1
echo "disc tool rc=$?"
tail -40 gpurun_out/r5_disc_parity.txt
