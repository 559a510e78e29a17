#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in 2 4; do python bench.py --global-views $V --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('global-views $V', 'ms/step', round(d['ms_per_step'],4), 'views/s', round(d['value'],1))"; done
