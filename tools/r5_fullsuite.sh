#!/bin/bash
# the whole -m gpu suite as the driver runs it + smoke (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu ) > gpurun_out/r5_full_pytest.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r5_full_pytest.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
