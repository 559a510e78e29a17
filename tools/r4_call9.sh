#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu ) > gpurun_out/r4c9_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4c9_pytest.log | cut -c1-300
