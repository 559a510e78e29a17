#!/bin/bash
# round-6 artefact set (GPU box): tools/profile_round.sh (rocprofv3 kernel trace + the three PMC passes + default / pipelined / C5 bench lines;
# C5 at EIGHT views per batch this round), a per-(kernel, grid) summary of the trace (the passes of the two sorts share kernels), the
# one-view-per-step shape with its kernel trace, and the same over a 1-rank RCCL communicator through include/dvs_comm.h
TAG=${1:-r06}
cd "$(dirname "$0")/.."
export C5_VIEWS=8
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
tail -18 gpurun_out/${TAG}_profile_round.log | cut -c1-160
bash tools/profile_single.sh $TAG > gpurun_out/${TAG}_profile_single.log 2>&1
DVS_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --global-views 1 --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_1view_rccl_1rank.json 2> gpurun_out/${TAG}_bench_1view_rccl_1rank.err
DVS_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 python bench.py --global-views 1 --a9-chunks 4 --no-cpu-baseline --profile-iters 0 --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_1view_rccl_1rank_chunks4.json 2> gpurun_out/${TAG}_bench_1view_rccl_1rank_chunks4.err
python - <<PY
import json
for f in ("${TAG}_bench.json", "${TAG}_single_bench.json", "${TAG}_bench_1view_rccl_1rank.json", "${TAG}_bench_1view_rccl_1rank_chunks4.json", "${TAG}_bench_c5.json", "${TAG}_bench_pipelined.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/" + f).read().splitlines() if l.startswith("{")][-1])
        print(f, "views/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "comm", d.get("t_comm_exposed_ms_per_step"), "nranks", d.get("rccl_nranks"))
    except Exception as e:
        print(f, "FAILED", e)
PY
