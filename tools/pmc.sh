#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   (run on the GPU box; writes gpurun_out/pmc_<tag>.txt)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $R/gpurun_out/pmc_$TAG
cd $R
rocprofv3 --kernel-trace --pmc $@ -d $R/gpurun_out/pmc_$TAG -o p -- python bench.py --steps 3 --warmup 1 --profile-iters 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_$TAG/stderr.log
python tools/rocpd_summary.py $R/gpurun_out/pmc_$TAG/p_results.db > $R/gpurun_out/pmc_$TAG.txt 2>&1
# keep db for inspection
grep -E 'k_render|k_preprocess' $R/gpurun_out/pmc_$TAG.txt
