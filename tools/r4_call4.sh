#!/bin/bash
# round-4 GPU call 4: parity of the A8 variants (prefetch of the next batch's records; owner-byte table update), their same-box A/B in the
# default 8-view step, and the remaining ablations (prologue only; staging only)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in pf own pfown; do
  DVS_RASTER_LIB=$PWD/tools/xlib/lib_$L.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_parity or live_lists or multi_view_batch" > gpurun_out/r4c4_parity_$L.log 2>&1
  echo "parity $L rc=$? $(tail -1 gpurun_out/r4c4_parity_$L.log)"
done
echo "== A/B"
ROUNDS=2 bash tools/r3_ab.sh tools/xlib/lib_exp.so tools/xlib/lib_pf.so tools/xlib/lib_own.so tools/xlib/lib_pfown.so 2>&1 | tee gpurun_out/r4c4_ab.txt
echo "== ablations (lib_exp): 64 = prologue only, 136 = no list loop and no publish (staging + prologue), 8 = no list loop"
export DVS_RASTER_LIB=$PWD/tools/xlib/lib_exp.so
for V in 64 136 8; do
env DVS_TR_DEBUG=$V timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-iters 3 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('DVS_TR_DEBUG=$V', 'ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"
done 2>&1 | tee gpurun_out/r4c4_a8_ablation2.txt
