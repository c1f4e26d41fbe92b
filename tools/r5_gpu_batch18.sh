#!/bin/bash
# round 5, GPU call 18: the two new full-size data-shaped tests (headline configuration on the disc scene; 2000^2 with three channels)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest "tests/test_gpu_fullsize.py::test_headline_r1_5000_disc_scene_matches_oracle" "tests/test_gpu_fullsize.py::test_config1_2000_three_channels_disc_scene_matches_oracle" -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b18_tests.txt 2>&1
echo "tests rc=$?"; grep "PARITY-FULLSIZE" gpurun_out/r5_b18_tests.txt | cut -c1-500; tail -2 gpurun_out/r5_b18_tests.txt
