#!/bin/bash
# one-view step: default sort passes vs the chained-scan variant
cd "$(dirname "$0")/.."
for i in 1 2; do for os in 0 1; do
DVS_SORT_ONESWEEP=$os python bench.py --global-views 1 --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('onesweep $os  ms/step', round(d['ms_per_step'],4))"
done; done
