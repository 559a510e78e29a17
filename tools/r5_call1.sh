#!/bin/bash
# round 5, GPU call 1: the segmented front end against the legacy one (bit-equality + stage times), then the parity tests that cover binning
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 > gpurun_out/r5_ab_v8.txt 2>&1; echo "ab v8 rc=$?"; tail -12 gpurun_out/r5_ab_v8.txt | cut -c1-700
timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 > gpurun_out/r5_ab_v1.txt 2>&1; echo "ab v1 rc=$?"; tail -8 gpurun_out/r5_ab_v1.txt | cut -c1-700
timeout 300 python tools/r5_ab.py 30000 256 200 3 3 > gpurun_out/r5_ab_small.txt 2>&1; echo "ab small rc=$?"; tail -6 gpurun_out/r5_ab_small.txt | cut -c1-400
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline_parity or multi_view or async or sort or tight or graph or golden" ) > gpurun_out/r5_c1_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r5_c1_pytest.log | cut -c1-300
