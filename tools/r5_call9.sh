#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for L in a9w0 a9w3 a9w4; do for NH in 0 1; do
echo "== lib_$L nohoist=$NH"; DVS_RASTER_LIB=$PWD/tools/xlib/lib_$L.so DVS_A9V_NOHOIST=$NH timeout 300 python tools/r5_ab.py 1000000 1920 1080 8 8 2>&1 | grep -v amdgpu.ids | cut -c1-420
done; done
echo "== release lib"; timeout 300 python tools/r5_ab.py 1000000 1920 1080 8 8 2>&1 | grep -v amdgpu.ids | cut -c1-420
echo "== V=1"; for L in a9w0 a9w3; do DVS_RASTER_LIB=$PWD/tools/xlib/lib_$L.so DVS_A9V_NOHOIST=1 timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 8 2>&1 | grep -v amdgpu.ids | cut -c1-420; done
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_multi_view_batch_equals_single_views" "tests/test_gpu_parity.py::test_depth_sort_digit_width_follows_the_key_range" -q -m gpu 2>&1 | tail -3
