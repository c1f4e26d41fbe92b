#!/bin/bash
# round 5, GPU call 2: the changed tests (disc parity regimes, pix2uv builds, self-check, one-shot cache, new full-size cases) + solo-band timing
mkdir -p gpurun_out
python -m pytest tests/test_gpu_disc.py tests/test_gpu_pix2uv.py tests/test_host_abi.py tests/test_gpu_oneshot.py tests/test_gpu_tiled.py tests/test_gpu_tiled_mp.py \
    -m gpu -q -rP -p no:cacheprovider -x > gpurun_out/r5_b2_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5_b2_tests.txt
python -m pytest "tests/test_gpu_fullsize.py::test_r2_parameter_set_at_5000_matches_oracle" "tests/test_gpu_fullsize.py::test_config3_quarter_scale_disc_scene_four_bands_and_plain_match_oracle" \
    -m gpu -q -rP -p no:cacheprovider > gpurun_out/r5_b2_fullsize.txt 2>&1
echo "fullsize rc=$?"; grep PARITY-FULLSIZE gpurun_out/r5_b2_fullsize.txt; tail -3 gpurun_out/r5_b2_fullsize.txt
export OCTANE_LIB=$PWD/octane_amd/liboctane_vof_diag.so
timeout -k 10 600 python tools/solo_band.py 10848 8 3 30 2,4,8 > gpurun_out/r5_solo_band_10848.txt 2>&1
echo "solo 10848 rc=$?"; cat gpurun_out/r5_solo_band_10848.txt
timeout -k 10 300 python tools/solo_band.py 5000 8 3 30 2,4,8 > gpurun_out/r5_solo_band_5000.txt 2>&1
echo "solo 5000 rc=$?"; cat gpurun_out/r5_solo_band_5000.txt
