#!/bin/bash
# round-4 artefact set (GPU box): tools/profile_round.sh (rocprofv3 kernel trace + the three PMC passes + default / pipelined / C5 bench lines),
# the one-view-per-step shape with its kernel trace, the same over a 1-rank RCCL communicator, and the opt-in tight-tiles line
TAG=${1:-r04}
cd "$(dirname "$0")/.."
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
tail -18 gpurun_out/${TAG}_profile_round.log | cut -c1-160
bash tools/profile_single.sh $TAG > gpurun_out/${TAG}_profile_single.log 2>&1
DVS_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --global-views 1 --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_1view_rccl_1rank.json 2> gpurun_out/${TAG}_bench_1view_rccl_1rank.err
python bench.py --tight-tiles 1 --no-cpu-baseline --steps 100 > gpurun_out/${TAG}_bench_tight_tiles.json 2> gpurun_out/${TAG}_bench_tight_tiles.err
python - <<PY
import json
for f in ("${TAG}_bench.json", "${TAG}_single_bench.json", "${TAG}_bench_1view_rccl_1rank.json", "${TAG}_bench_c5.json", "${TAG}_bench_pipelined.json", "${TAG}_bench_tight_tiles.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/" + f).read().splitlines() if l.startswith("{")][-1])
        print(f, "views/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "comm", d.get("t_comm_exposed_ms_per_step"), "other", d.get("other_grad_mode"))
    except Exception as e:
        print(f, "FAILED", e)
PY
