#!/bin/bash
# round-4 final check (GPU box): the whole -m gpu suite as the driver runs it, smoke(), then the artefact set
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r4f_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4f_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/r4_profile.sh r04
