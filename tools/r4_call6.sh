#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tight or onesweep" > gpurun_out/r4c6_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4c6_pytest.log | cut -c1-300
echo "== canonical vs tight tiles (release lib)"
for i in 1 2; do for TT in 0 1; do
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-iters 3 --tight-tiles $TT 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
ks={k['stage']: round(k['ms_per_launch_set'],3) for k in d['roofline'].get('kernels', [])}
print('tight=$TT', 'views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'T', d['config']['T'], ks)"
done; done 2>&1 | tee gpurun_out/r4c6_tight_ab.txt
echo "== A8 staging ablation (lib_exp): 136 = prologue + staging; 16520 = the same without the ellipse-vs-block tests; 64 = prologue only"
export DVS_RASTER_LIB=$PWD/tools/xlib/lib_exp.so
for V in 136 16520 64 136 16520; do
env DVS_TR_DEBUG=$V timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-iters 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('DVS_TR_DEBUG=$V', 'ms/step', round(d['ms_per_step'],4))"
done 2>&1 | tee gpurun_out/r4c6_a8_staging_ablation.txt
