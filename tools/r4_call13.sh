#!/bin/bash
# round-4 GPU call 13: wave-autonomous A8 (-DTR_AUTONOMOUS=1: one wave per workgroup, no workgroup barrier) — parity, then same-box A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in auto; do
  DVS_RASTER_LIB=$PWD/tools/xlib/lib_$L.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_parity or live_lists or multi_view_batch" > gpurun_out/r4c13_parity_$L.log 2>&1
  echo "parity $L rc=$? $(tail -1 gpurun_out/r4c13_parity_$L.log)"
done
echo "== A/B"
ROUNDS=2 bash tools/r3_ab.sh tools/xlib/lib_exp.so tools/xlib/lib_auto.so tools/xlib/lib_auto4.so 2>&1 | tee gpurun_out/r4c13_ab.txt
echo "== one view per step"
for L in exp auto; do
DVS_RASTER_LIB=$PWD/tools/xlib/lib_$L.so timeout 300 python bench.py --global-views 1 --steps 200 --warmup 20 --no-cpu-baseline --profile-iters 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('1-view $L views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4))"
done 2>&1 | tee -a gpurun_out/r4c13_ab.txt
