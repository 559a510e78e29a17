#!/bin/bash
# round 6 (GPU box): the inputs of tools/valu_ceiling.py — two SQ class-counter passes of the default bench command, the issue-cost micro-benchmark
# and its clock. Writes gpurun_out/r06_valu_ceiling.json (+ the raw passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R; mkdir -p gpurun_out
( cd /tmp && bash $R/tools/pmc.sh r06_valu_a SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 GRBM_GUI_ACTIVE ) > /dev/null 2>&1
( cd /tmp && bash $R/tools/pmc.sh r06_valu_b SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE ) > /dev/null 2>&1
[ -x tools/ubench/valu_cost ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_cost.hip -o tools/ubench/valu_cost
tools/ubench/valu_cost > gpurun_out/r06_valu_cost.txt 2>&1
rm -rf gpurun_out/pmc_r06_ubench; ( cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_r06_ubench -o p -- $R/tools/ubench/valu_cost > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/pmc_r06_ubench/p_results.db > gpurun_out/pmc_r06_ubench.txt 2>&1
python tools/valu_ceiling.py gpurun_out/pmc_r06_valu_a.txt gpurun_out/pmc_r06_valu_b.txt gpurun_out/r06_valu_cost.txt gpurun_out/pmc_r06_ubench.txt profiles/r06_valu_static_mix.json > gpurun_out/r06_valu_ceiling.json
cat gpurun_out/r06_valu_ceiling.json | head -80; cat gpurun_out/r06_valu_cost.txt | cut -c1-100
