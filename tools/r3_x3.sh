#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_lineage_mode.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/bwd_probe.py --reps 20 t64:variant=tr64,fwd=quadrant t64b:variant=tr64,fwd=quadrant blocks:variant=blocks,fwd=quadrant noloop:variant=tr64,fwd=quadrant,DVS_TR_DEBUG=8 noflush:variant=tr64,fwd=quadrant,DVS_TR_DEBUG=4 noatom:variant=tr64,fwd=quadrant,DVS_TR_DEBUG=1 2>&1 | grep -v amdgpu.ids
for v in tr64; do
  timeout 300 python bench.py --steps 30 --warmup 5 --bwd-variant $v --no-cpu-baseline --profile-iters 3 > gpurun_out/a8_bench_$v.json 2> gpurun_out/a8_bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/a8_bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "views/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "A8 ms/launch", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("$v bench failed", e); print(open("gpurun_out/a8_bench_$v.err").read()[-2000:])
PY
done
