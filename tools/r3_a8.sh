#!/bin/bash
# round 3: parity + timing of the A8 "tr" variants against "blocks" (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_lineage_mode.py -m gpu -x -q > gpurun_out/a8_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/a8_pytest.log
timeout 300 python tools/bwd_probe.py --reps 20 blocks:variant=blocks,fwd=quadrant tr:variant=tr,fwd=quadrant tr64:variant=tr64,fwd=quadrant > gpurun_out/a8_probe.txt 2>&1
cat gpurun_out/a8_probe.txt
for v in blocks tr tr64; do
  timeout 300 python bench.py --steps 30 --warmup 5 --bwd-variant $v --no-cpu-baseline --profile-iters 3 > gpurun_out/a8_bench_$v.json 2> gpurun_out/a8_bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/a8_bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "views/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "A8 ms/launch", d["roofline"]["avg_launch_ms"], "sclk", d["clocks"])
except Exception as e:
    print("$v bench failed", e); print(open("gpurun_out/a8_bench_$v.err").read()[-2000:])
PY
done
