#!/bin/bash
# round-4 GPU call 11: the retired A8 variants (mm, reduce) through the replay-based parity matrix; the plugin with DVS_TIGHT_TILES=1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
DVS_TEST_ALL_VARIANTS=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_lineage_mode.py -q -m gpu -k "pipeline_parity or multi_view_batch or lineage" > gpurun_out/r4c11_pytest.log 2>&1; echo "all-variants pytest rc=$? $(tail -1 gpurun_out/r4c11_pytest.log)"
for TT in 0 1; do
DVS_TIGHT_TILES=$TT timeout 300 divshot_amd/lib/gaussian_train --inputPath synthetic:N=200000,W=640,H=480,cams=6,sh=2,seed=3 --maxIteration 600 --densifyStrategy 0 --warmupLength 100 --refineEvery 100 --refineStopIter 450 --outputPath gpurun_out/tt$TT/it > gpurun_out/r4c11_cli_tt$TT.log 2>&1
echo "tight=$TT rc=$? $(grep -c 'densify @' gpurun_out/r4c11_cli_tt$TT.log) refinements; $(grep 'Iteraions 500' gpurun_out/r4c11_cli_tt$TT.log | tail -1); $(grep -o 'train step : 500/600.*' gpurun_out/r4c11_cli_tt$TT.log | tail -1)"
grep "densify @" gpurun_out/r4c11_cli_tt$TT.log | tr '\n' ';'; echo
done
rm -rf gpurun_out/tt0 gpurun_out/tt1
