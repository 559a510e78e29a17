#!/bin/bash
# round 5, GPU call 2: front-end A/B after the rowscan / DPP / padded-counter changes, per-kernel trace of the segmented pass, parity subset
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 > gpurun_out/r5_ab_v8.txt 2>&1; echo "ab v8 rc=$?"; grep -A1 "^\[" gpurun_out/r5_ab_v8.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v8.txt
timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 > gpurun_out/r5_ab_v1.txt 2>&1; echo "ab v1 rc=$?"; grep -A1 "^\[" gpurun_out/r5_ab_v1.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v1.txt
rm -rf gpurun_out/prof_r5c2
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5c2 -o p -- python $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 --profile-iters 0 > $R/gpurun_out/r5c2_bench_under_rocprof.json 2> $R/gpurun_out/r5c2_bench.stderr )
python tools/rocpd_summary.py gpurun_out/prof_r5c2/p_results.db --by-grid > gpurun_out/r5c2_kernel_stats.txt 2>&1; head -30 gpurun_out/r5c2_kernel_stats.txt | cut -c1-150
rm -rf gpurun_out/prof_r5c2
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_lineage_mode.py tests/test_abi.py -x -q -m gpu ) > gpurun_out/r5_c2_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r5_c2_pytest.log | cut -c1-300
