#!/bin/bash
# usage (GPU box): tools/profile_plugin.sh <tag> -> gpurun_out/<tag>_plugin_kernel_stats.txt + <tag>_plugin.log
# rocprofv3 kernel trace of libgstrain's full training loop through the CLI host: 3300 iterations, MCMC growing 1M -> 3M splats at 1080p.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/prof_plugin_$TAG
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_plugin_$TAG -o p -- $R/divshot_amd/lib/gaussian_train \
  --inputPath synthetic:N=1000000,W=1920,H=1080,cams=8,sh=3,seed=1 --maxIteration 3300 --outputPath /tmp/plugin_out/iteration \
  --densifyStrategy 1 --warmupLength 500 --refineEvery 100 > $R/gpurun_out/${TAG}_plugin.log 2>&1
cd $R && python tools/rocpd_summary.py gpurun_out/prof_plugin_$TAG/p_results.db > gpurun_out/${TAG}_plugin_kernel_stats.txt 2>&1
grep -E "Iteraions (0|1000|2000|3000|3200)," gpurun_out/${TAG}_plugin.log | tail -5; grep -E "mcmc @" gpurun_out/${TAG}_plugin.log | tail -2; tail -2 gpurun_out/${TAG}_plugin.log
head -14 gpurun_out/${TAG}_plugin_kernel_stats.txt
