#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline_parity" > gpurun_out/a8_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/a8_pytest.log
timeout 300 python tools/bwd_probe.py --reps 20 blocks:variant=blocks,fwd=quadrant blocks6:variant=blocks,fwd=quadrant,DVS_BWD_EXTRA_LDS=7000 blocks5:variant=blocks,fwd=quadrant,DVS_BWD_EXTRA_LDS=11000 tr:variant=tr,fwd=quadrant tr64:variant=tr64,fwd=quadrant > gpurun_out/a8_probe.txt 2>&1
cat gpurun_out/a8_probe.txt
(cd /tmp && rocprofv3 -L > $OLDPWD/gpurun_out/counters.txt 2>&1)
for v in blocks tr64; do
bash tools/pmc_probe.sh a8_${v}_1 ${v}:variant=${v},fwd=quadrant SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
bash tools/pmc_probe.sh a8_${v}_2 ${v}:variant=${v},fwd=quadrant SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_SMEM
done
