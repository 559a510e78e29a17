#!/bin/bash
# round-4 GPU call 14: occupancy sensitivity of A8 inside the default 8-view step (experiment knob DVS_BWD_EXTRA_LDS: dynamic LDS padding)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DVS_RASTER_LIB=$PWD/tools/xlib/lib_exp.so
for X in 0 4000 6000 14000 0; do
env DVS_BWD_EXTRA_LDS=$X timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-iters 3 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('DVS_BWD_EXTRA_LDS=$X (LDS per workgroup', 26496+$X, 'B) ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"
done 2>&1 | tee gpurun_out/r4c14_occupancy.txt
