#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
rm -rf gpurun_out/prof_r5c3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5c3 -o p -- python $R/tools/r5_sortbench.py 1000000 8000000 > $R/gpurun_out/r5c3_sortbench.txt 2>&1 )
cat gpurun_out/r5c3_sortbench.txt | grep -v amdgpu.ids
python tools/rocpd_summary.py gpurun_out/prof_r5c3/p_results.db --by-grid > gpurun_out/r5c3_kernel_stats.txt 2>&1; grep -E "k_seg|k_sort" gpurun_out/r5c3_kernel_stats.txt | cut -c1-150
rm -rf gpurun_out/prof_r5c3
