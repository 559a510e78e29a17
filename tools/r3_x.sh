#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tools/bwd_probe.py --reps 20 "$@" > gpurun_out/x_probe.txt 2>&1
cat gpurun_out/x_probe.txt | grep -v amdgpu.ids
