#!/bin/bash
# round-4 GPU call 2: remaining -m gpu tests, then the A8 ablations INSIDE the default 8-view step (experiment build; timing only)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_large.py --deselect tests/test_golden.py > gpurun_out/r4c2_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4c2_pytest.log | cut -c1-300
for f in train_step_parity two_rank_plugin; do [ -f gpurun_out/$f.json ] && { echo $f; cat gpurun_out/$f.json; echo; }; done
echo "== A8 ablations in the 8-view step (lib_exp)"
export DVS_RASTER_LIB=$PWD/tools/xlib/lib_exp.so
for V in 0 4 512 1024 2048 4096 8192 1 128 8 0; do
env DVS_TR_DEBUG=$V timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-iters 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('DVS_TR_DEBUG=$V', 'ms/step', round(d['ms_per_step'],4), 'A8', round(d['roofline']['avg_launch_ms'],4))"
done 2>&1 | tee gpurun_out/r4c2_a8_ablation_8view.txt
