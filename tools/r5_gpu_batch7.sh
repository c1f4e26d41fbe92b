#!/bin/bash
# round 5, GPU call 7: IPC open times (fixed probe), solo-band timing with per-level GPU times
mkdir -p gpurun_out
timeout -k 10 400 python tools/ipc_probe.py 1 4 8 16 19 > gpurun_out/r5_ipc_probe.txt 2>&1
echo "ipc probe rc=$?"; cat gpurun_out/r5_ipc_probe.txt | grep -v amdgpu.ids
export OCTANE_LIB=$PWD/octane_amd/liboctane_vof_diag.so
timeout -k 10 600 python tools/solo_band.py 10848 8 3 30 2,4,8 > gpurun_out/r5_solo_band_10848.txt 2>&1
echo "solo 10848 rc=$?"; cut -c1-700 gpurun_out/r5_solo_band_10848.txt
timeout -k 10 300 python tools/solo_band.py 5000 8 3 30 2,4,8 > gpurun_out/r5_solo_band_5000.txt 2>&1
echo "solo 5000 rc=$?"; cut -c1-700 gpurun_out/r5_solo_band_5000.txt
