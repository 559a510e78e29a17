#!/bin/bash
# usage (GPU box): tools/profile_single.sh <tag> -> gpurun_out/<tag>_single_kernel_stats.txt: kernel trace of one view per step (the N=8 per-rank shape)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
rm -rf $R/gpurun_out/prof_single_$TAG
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_single_$TAG -o p -- python $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 --profile-iters 0 --global-views 1 > $R/gpurun_out/${TAG}_single_bench.json 2>/dev/null
cd $R && python tools/rocpd_summary.py gpurun_out/prof_single_$TAG/p_results.db > gpurun_out/${TAG}_single_kernel_stats.txt 2>&1
head -24 gpurun_out/${TAG}_single_kernel_stats.txt | cut -c1-125
