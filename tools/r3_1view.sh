#!/bin/bash
cd "$(dirname "$0")/.."
for ch in 1 2 4; do
DVS_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$ch python bench.py --global-views 1 --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 --a9-chunks $ch 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('chunks', d['config']['a9_chunks'], 'ms/step', round(d['ms_per_step'],4), 'comm exposed', d['t_comm_exposed_ms_per_step']['mean'])"
done
python bench.py --global-views 1 --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('plain ms/step', round(d['ms_per_step'],4))"
