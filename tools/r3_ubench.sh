#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/ubench/lds_atomic.hip -o /tmp/lds_atomic 2>/dev/null && /tmp/lds_atomic > gpurun_out/lds_atomic.txt 2>&1
cat gpurun_out/lds_atomic.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -S --cuda-device-only tools/ubench/lds_atomic.hip -o /tmp/lds_atomic.s 2>/dev/null; grep -n "ds_add\|ds_cmpst\|ds_pk" /tmp/lds_atomic.s | head
