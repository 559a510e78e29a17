#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>     e.g. r02
# Produces under gpurun_out/: <tag>_bench.json (the default command), <tag>_bench_pipelined.json (the round-1 shape: one view per pass,
# two streams), <tag>_bench_c5.json, <tag>_bench_under_rocprof.json + <tag>_kernel_stats.txt (rocprofv3 --kernel-trace --stats of the
# default bench command), pmc_<tag>_sq.txt (SQ counters, own pass), pmc_<tag>_fetch.txt / pmc_<tag>_write.txt (FETCH_SIZE / WRITE_SIZE,
# own passes) and <tag>_traffic.json built from them. Copy what should be judged into profiles/ afterwards.
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${TAG}
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o p -- python $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 --profile-iters 0 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_bench.stderr )
python tools/rocpd_summary.py gpurun_out/prof_${TAG}/p_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python tools/rocpd_summary.py gpurun_out/prof_${TAG}/p_results.db --by-grid > gpurun_out/${TAG}_kernel_stats_by_grid.txt 2>&1
( cd /tmp && bash $R/tools/pmc.sh ${TAG}_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE ) > /dev/null 2>&1
( cd /tmp && bash $R/tools/pmc.sh ${TAG}_fetch FETCH_SIZE ) > /dev/null 2>&1
( cd /tmp && bash $R/tools/pmc.sh ${TAG}_write WRITE_SIZE ) > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/pmc_${TAG}_fetch.txt gpurun_out/pmc_${TAG}_write.txt 4 > gpurun_out/${TAG}_traffic.json 2>/dev/null
# bench.py reads the latest profiles/r*_traffic.json and r*_pmc_sq.txt for roofline.traffic / valu_issue: make this round's visible to it
cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json; cp gpurun_out/pmc_${TAG}_sq.txt profiles/${TAG}_pmc_sq.txt
python bench.py > gpurun_out/${TAG}_bench.json 2>> gpurun_out/${TAG}_bench.stderr
python bench.py --mode pipelined --no-cpu-baseline --steps 100 > gpurun_out/${TAG}_bench_pipelined.json 2>> gpurun_out/${TAG}_bench.stderr
python bench.py --workload C5 --global-views ${C5_VIEWS:-4} --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_c5.json 2>> gpurun_out/${TAG}_bench.stderr
head -c 400 gpurun_out/${TAG}_bench.json; echo
head -16 gpurun_out/${TAG}_kernel_stats.txt
