#!/bin/bash
# usage (GPU box): tools/pmc_probe.sh <tag> <probe-config> <counters...>  -> gpurun_out/pmc_<tag>.txt (PMC pass over tools/bwd_probe.py)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CFG=$2; shift; shift
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
rocprofv3 --kernel-trace --pmc $@ -d $R/gpurun_out/pmc_$TAG -o p -- python $R/tools/bwd_probe.py --reps 5 $CFG > $R/gpurun_out/pmc_$TAG/stdout.log 2> $R/gpurun_out/pmc_$TAG/stderr.log
cd $R
python tools/rocpd_summary.py $R/gpurun_out/pmc_$TAG/p_results.db > $R/gpurun_out/pmc_$TAG.txt 2>&1
grep -E 'k_render_bwd' $R/gpurun_out/pmc_$TAG.txt
