#!/bin/bash
# full GPU check of the tree (GPU box): pytest -m gpu, smoke(), the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py ${BENCH_ARGS:---steps 50 --warmup 10} > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
    print("views/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), "A8 ms", round(d["roofline"]["avg_launch_ms"],3), "e2e frac", round(d["pipeline"]["frac_of_hbm_peak_end_to_end"],4), "strict", round(d["strict_single_view"]["ms_per_view"],3))
    print("parity", d.get("parity_vs_oracle"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_default.err").read()[-3000:])
PY
