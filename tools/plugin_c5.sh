#!/bin/bash
# BASELINE config C5's shape through the plugin (GPU box): 5 M splats at 3840x2160, SH degree 3, ADC active, two refinements; the log
# (stderr: config, loss lines, `densify @`, `raster @`: T, instance arena, growth events, overflowed forwards) -> gpurun_out/<tag>_plugin_c5.log
TAG=${1:-r05}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/c5run
( time timeout 1500 divshot_amd/lib/gaussian_train --inputPath synthetic:N=5000000,W=3840,H=2160,cams=8,sh=3,seed=2 --maxIteration 50 --outputPath /tmp/c5run/iteration \
  --warmupLength 5 --refineEvery 20 --refineStopIter 45 --densifyStrategy 0 --progressTrain 0 --ssim 0.2 ) > gpurun_out/${TAG}_plugin_c5.stdout 2> gpurun_out/${TAG}_plugin_c5.log
echo "rc=$?"; grep -E "densify @|raster @|Iteraions|config:|IGNORED|error|Error|CAPACITY" gpurun_out/${TAG}_plugin_c5.log | cut -c1-260 | head -30; tail -3 gpurun_out/${TAG}_plugin_c5.stdout
