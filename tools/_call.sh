#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_into" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"; done
