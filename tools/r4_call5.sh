#!/bin/bash
# round-4 GPU call 5: full -m gpu suite (v_cmpx forward as default, DVS_TILES_TIGHT, everything else), smoke, canonical vs tight-tiles bench,
# A7 staging-only ablation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r4c5_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r4c5_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== canonical vs tight tiles (release lib)"
for i in 1 2; do for TT in 0 1; do
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-iters 3 --tight-tiles $TT 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
ks={k['stage']: round(k['ms_per_launch_set'],3) for k in d['roofline'].get('kernels', [])}
print('tight=$TT', 'views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'T', d['config']['T'], ks)"
done; done 2>&1 | tee gpurun_out/r4c5_tight_ab.txt
echo "== A7 staging only (lib_exp, DVS_FWD_DEBUG=1) vs base"
for V in 0 1; do
DVS_RASTER_LIB=$PWD/tools/xlib/lib_exp.so DVS_FWD_DEBUG=$V timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --profile-iters 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('DVS_FWD_DEBUG=$V', 'ms/step', round(d['ms_per_step'],4))"
done 2>&1 | tee gpurun_out/r4c5_a7_ablation.txt
