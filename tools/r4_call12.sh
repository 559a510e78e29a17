#!/bin/bash
# round-4 GPU call 12: the opt-in tight tile bounds in the one-view-per-step shape (the per-rank shape of an 8-GPU run) and in the plugin
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do for TT in 0 1; do
timeout 300 python bench.py --global-views 1 --steps 200 --warmup 20 --no-cpu-baseline --profile-iters 0 --tight-tiles $TT 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('1-view tight=$TT views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'T', d['config']['T'])"
done; done 2>&1 | tee gpurun_out/r4c12_tight_1view.txt
for i in 1 2; do for TT in 0 1; do
DVS_TIGHT_TILES=$TT timeout 300 divshot_amd/lib/gaussian_train --inputPath synthetic:N=1000000,W=1920,H=1080,cams=8,sh=3,seed=1 --maxIteration 600 --densifyStrategy 0 --warmupLength 100000 --outputPath gpurun_out/tt$TT/it 2>&1 | grep -o 'train step : 500/600.*' | sed "s/^/plugin C3 tight=$TT /"
done; done 2>&1 | tee gpurun_out/r4c12_tight_plugin.txt
rm -rf gpurun_out/tt0 gpurun_out/tt1
