#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest "tests/test_train_step.py::test_plugin_trajectory_on_the_cli_defaults_ssim_sh3_eight_views" "tests/test_gpu_parity.py::test_depth_sort_digit_width_follows_the_key_range" -q -m gpu ) > gpurun_out/r5_c8_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5_c8_pytest.log | cut -c1-400
bash tools/r5_profile.sh r05a
