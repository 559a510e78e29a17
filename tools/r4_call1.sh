#!/bin/bash
# round-4 GPU call 1: the whole -m gpu suite (new: two-rank plugin over the TCP test backend, step-level train_step parity), then same-box
# A/B of the A7 v_cmpx visit and the A8 ablation upper bounds (experiment builds of tools/xbuild.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4c1_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4c1_pytest.log
for f in train_step_parity two_rank_plugin; do [ -f gpurun_out/$f.json ] && { echo $f; cat gpurun_out/$f.json; echo; }; done
echo "== A/B exp vs cmpx"
ROUNDS=2 bash tools/r3_ab.sh tools/xlib/lib_exp.so tools/xlib/lib_cmpx.so 2>&1 | tee gpurun_out/r4c1_ab.txt
echo "== A8 ablations (kernel alone, C3 one view)"
DVS_RASTER_LIB=$PWD/tools/xlib/lib_exp.so timeout 900 python tools/bwd_probe.py --reps 20 base:variant=tr,fwd=quadrant nobarrier:variant=tr,fwd=quadrant,DVS_TR_DEBUG=256 nopad:variant=tr,fwd=quadrant,DVS_TR_DEBUG=512 noarith:variant=tr,fwd=quadrant,DVS_TR_DEBUG=1024 noflush:variant=tr,fwd=quadrant,DVS_TR_DEBUG=4 nobar_nopad:variant=tr,fwd=quadrant,DVS_TR_DEBUG=768 base2:variant=tr,fwd=quadrant 2>&1 | tee gpurun_out/r4c1_a8_probe.txt
