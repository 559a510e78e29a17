#!/bin/bash
# same-box A/B of an environment switch on the default bench step: ROUNDS=3 tools/r3_env_ab.sh VAR val0 val1 [bench args...]
cd "$(dirname "$0")/.."
R=${ROUNDS:-3}; VAR=$1; A=$2; B=$3; shift 3
for i in $(seq 1 $R); do
for V in $A $B; do
env $VAR=$V timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-iters 3 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$VAR=$V', 'views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"
done; done
