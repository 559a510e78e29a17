#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
rm -rf gpurun_out/prof_r5c4
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5c4 -o p -- python $R/tools/r5_sortbench.py 1000000 8000000 > $R/gpurun_out/r5c4_sortbench.txt 2>&1 )
grep "^\[" gpurun_out/r5c4_sortbench.txt
python tools/rocpd_summary.py gpurun_out/prof_r5c4/p_results.db --by-grid > gpurun_out/r5c4_kernel_stats.txt 2>&1; grep -E "k_seg|k_sort" gpurun_out/r5c4_kernel_stats.txt | cut -c1-150
rm -rf gpurun_out/prof_r5c4
timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 > gpurun_out/r5_ab_v8.txt 2>&1; echo "ab v8 rc=$?"; grep -A1 "^\[" gpurun_out/r5_ab_v8.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v8.txt
DVS_FE_ITEMS=8 timeout 600 python tools/r5_ab.py 1000000 1920 1080 8 10 > gpurun_out/r5_ab_v8_items8.txt 2>&1; echo "ab v8 items8 rc=$?"; grep -A1 "^\[seg" gpurun_out/r5_ab_v8_items8.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v8_items8.txt
timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 > gpurun_out/r5_ab_v1.txt 2>&1; echo "ab v1 rc=$?"; grep -A1 "^\[" gpurun_out/r5_ab_v1.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v1.txt
DVS_FE_ITEMS=16 timeout 300 python tools/r5_ab.py 1000000 1920 1080 1 10 > gpurun_out/r5_ab_v1_items16.txt 2>&1; echo "ab v1 items16 rc=$?"; grep -A1 "^\[seg" gpurun_out/r5_ab_v1_items16.txt | cut -c1-420; tail -1 gpurun_out/r5_ab_v1_items16.txt
