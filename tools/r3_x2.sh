#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== spread build"; timeout 600 python tools/bwd_probe.py --reps 20 t64:variant=tr64,fwd=quadrant rounds:variant=tr64,fwd=quadrant,DVS_TR_DEBUG=16 notab:variant=tr64,fwd=quadrant,DVS_TR_DEBUG=2 2>&1 | grep -v amdgpu.ids
echo "== adjacent build"; DVS_RASTER_LIB=$PWD/tools/xlib/lib_nospread.so timeout 600 python tools/bwd_probe.py --reps 20 t64:variant=tr64,fwd=quadrant rounds:variant=tr64,fwd=quadrant,DVS_TR_DEBUG=16 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline_parity" 2>&1 | tail -3
