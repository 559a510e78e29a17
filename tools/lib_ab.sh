#!/bin/bash
# same-box A/B of builds of libdvsraster.so on the default bench step: ROUNDS=3 tools/lib_ab.sh libA.so libB.so ...
cd "$(dirname "$0")/.."
R=${ROUNDS:-3}
for i in $(seq 1 $R); do
for L in "$@"; do
DVS_RASTER_LIB=$PWD/$L timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-iters 3 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$L', 'views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"
done; done
