#!/bin/bash
# same-box A/B of two builds of libdvsraster.so on the default bench step: tools/r3_ab.sh libA.so libB.so [rounds]
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
for L in $A $B; do
DVS_RASTER_LIB=$PWD/$L timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-iters 3 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$L', 'views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"
done; done
