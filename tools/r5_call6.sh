#!/bin/bash
# A6 fusion A/B + tests that touch ranges / exported state + new train_step tests + the failed plugin test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 > gpurun_out/r5_ab_v8.txt 2>&1; echo "ab v8 rc=$?"; grep -A1 "^\[seg" gpurun_out/r5_ab_v8.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v8.txt
DVS_FE_NO_FUSE_A6=1 timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 > gpurun_out/r5_ab_v8_nofuse.txt 2>&1; echo "ab v8 nofuse rc=$?"; grep -A1 "^\[seg" gpurun_out/r5_ab_v8_nofuse.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v8_nofuse.txt
timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 > gpurun_out/r5_ab_v1.txt 2>&1; echo "ab v1 rc=$?"; grep -A1 "^\[seg" gpurun_out/r5_ab_v1.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v1.txt
timeout 300 python tools/r5_ab.py 30000 256 200 3 3 > gpurun_out/r5_ab_small.txt 2>&1; echo "ab small rc=$?"; tail -1 gpurun_out/r5_ab_small.txt
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_large.py tests/test_train_step.py "tests/test_plugin.py::test_cli_eight_views_per_iteration_as_one_pass" -q -m gpu -x ) > gpurun_out/r5_c6_pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r5_c6_pytest.log | cut -c1-600
