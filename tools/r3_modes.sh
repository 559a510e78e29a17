#!/bin/bash
# step time of the grouped / pipelined step shapes next to the default batch (same box)
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-iters 0 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$*', 'views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4))"; }
run --mode batch
run --mode grouped --groups 2 --contexts 2
run --mode grouped --groups 2 --contexts 2 --stagger 1
run --mode grouped --groups 4 --contexts 2
run --mode batch
