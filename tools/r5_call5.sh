#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_train_step.py tests/test_plugin.py tests/test_abi.py -q -m gpu -x ) > gpurun_out/r5_c5_pytest.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r5_c5_pytest.log | cut -c1-600
