#!/bin/bash
# round-4 GPU call 8: deferred table passes (TR_DEFER) at occupancy 5 and 6 — parity, then same-box A/B; m5 = the default code at occupancy 5
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in df5 df6; do
  DVS_RASTER_LIB=$PWD/tools/xlib/lib_$L.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_parity or live_lists or multi_view_batch" > gpurun_out/r4c8_parity_$L.log 2>&1
  echo "parity $L rc=$? $(tail -1 gpurun_out/r4c8_parity_$L.log)"
done
echo "== A/B"
ROUNDS=2 bash tools/r3_ab.sh tools/xlib/lib_exp.so tools/xlib/lib_df5.so tools/xlib/lib_df6.so tools/xlib/lib_m5.so 2>&1 | tee gpurun_out/r4c8_ab.txt
echo "== one view per step (the per-rank shape): default and hip graph"
for G in 0 1; do
timeout 300 python bench.py --global-views 1 --steps 200 --warmup 20 --no-cpu-baseline --profile-iters 0 --graph $G 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('graph=$G views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4))"
done 2>&1 | tee gpurun_out/r4c8_graph_1view.txt
