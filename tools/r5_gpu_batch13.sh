#!/bin/bash
# round 5, GPU call 13: BASELINE configs[3] at FULL size on the data-shaped (disc) scene against the oracle, plain plan and four row bands, once, as a record
mkdir -p gpurun_out
timeout -k 10 1100 python tools/runaway_check.py 10848 32544 4 disc > gpurun_out/r5_config3_fullsize_disc_vs_oracle.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r5_config3_fullsize_disc_vs_oracle.txt
