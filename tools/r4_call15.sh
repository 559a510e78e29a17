#!/bin/bash
# round-4 GPU call 15: timing probe of A8 at SEVEN workgroups per CU (tables shrunk to 8 floats per row — wrong results — so that the LDS
# footprint allows it; -DTR_MINW=7: 72 VGPRs, 4 spilled); tab8 = the same shrunk tables at six workgroups (the control)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
ROUNDS=2 bash tools/r3_ab.sh tools/xlib/lib_exp.so tools/xlib/lib_tab8.so tools/xlib/lib_occ7.so 2>&1 | tee gpurun_out/r4c15_occ7.txt
