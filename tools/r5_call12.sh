#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== key16"; timeout 300 python tools/r5_ab.py 1000000 1920 1080 8 10 2>&1 | grep -v amdgpu.ids | cut -c1-420
echo "== no key16"; DVS_FE_NO_KEY16=1 timeout 300 python tools/r5_ab.py 1000000 1920 1080 8 10 2>&1 | grep -v amdgpu.ids | cut -c1-420
echo "== key16"; timeout 300 python tools/r5_ab.py 1000000 1920 1080 8 10 2>&1 | grep -v amdgpu.ids | cut -c1-420
echo "== V=1 key16"; timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 2>&1 | grep -v amdgpu.ids | cut -c1-420
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -x -k "pipeline_parity or multi_view or async or sort or tight or golden or fused or digit" ) > gpurun_out/r5_c12_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5_c12_pytest.log | cut -c1-300
