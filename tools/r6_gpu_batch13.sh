#!/bin/bash
# round 6, GPU call 13: BASELINE configs[3] at FULL size (10848^2, data-shaped disc scene) on the final kernels -- default plan, plan without the
# persistent solve (diagnostic library), four row bands -- against the CPU oracle (~6 minutes of oracle on 16 cores)
mkdir -p gpurun_out
timeout -k 10 1100 python tools/runaway_check.py 10848 32544 4 disc > gpurun_out/r6_config3_fullsize_disc_vs_oracle.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r6_config3_fullsize_disc_vs_oracle.txt
