#!/bin/bash
# round 5, GPU call 11: what forming u + x + a p inside k_assemble would cost (variant built out of tree), and the single-band placement-trials test
mkdir -p gpurun_out
timeout -k 10 400 python tools/experiments/asm_fuse_probe.py time 5000 2500 1250 > gpurun_out/r5_asm_fuse_probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r5_asm_fuse_probe.txt
timeout -k 10 300 python -m pytest tests/test_gpu_tiled.py -m gpu -q -p no:cacheprovider -k "one_band" 2>&1 | tail -2
