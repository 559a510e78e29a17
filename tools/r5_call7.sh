#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_train_step.py tests/test_abi.py "tests/test_gpu_parity.py::test_depth_sort_digit_width_follows_the_key_range" "tests/test_gpu_parity.py::test_tile_ranges_fused_into_the_sort_equal_the_separate_kernel" "tests/test_gpu_parity.py::test_sort_pairs" "tests/test_gpu_parity.py::test_async_forward_no_host_sync" "tests/test_gpu_parity.py::test_multi_view_batch_equals_single_views" tests/test_golden.py -q -m gpu ) > gpurun_out/r5_c7_pytest.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r5_c7_pytest.log | cut -c1-500
timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 2>&1 | grep -v amdgpu.ids | cut -c1-420
timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 2>&1 | grep -v amdgpu.ids | cut -c1-420
