#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/quadrant_stats.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_quadrant_stats.json
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_depth_sort_digit_width_follows_the_key_range" -q -m gpu 2>&1 | tail -2
