#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_golden.py -q -m gpu -k "pipeline_parity or onesweep or full_size or golden or edge" ) > gpurun_out/r4c10_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4c10_pytest.log | cut -c1-300
python - <<'PY'
import json
for l in open("gpurun_out/parity_report.jsonl"):
    r = json.loads(l)
    worst = max(v.get("clean_worst_err_over_tol", 0) for run in r["runs"].values() for v in run.values())
    print(r["config"], "tainted", round(r["tainted_splat_fraction"], 4), "replay", r["decision_replay"], "worst clean err/tol", round(worst, 3))
PY
