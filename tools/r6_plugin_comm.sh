#!/bin/bash
# round 6 (GPU box): libgstrain's full training iteration at C3 (1M splats, 1080p, one view per step = the per-rank shape of an 8-GPU run)
# plain, and with every collective of the data-parallel step executed by RCCL on a 1-rank communicator (DVS_FORCE_COMM=1: the identity, so
# what is measured is what the exchange's launches, events and stream hops cost): unchunked / 4 A9 chunks / 4 chunks PIPELINED across the
# iteration boundary (DVS_EXCHANGE_PIPELINE=1). it/s from the host's own progress line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS="--inputPath synthetic:N=1000000,W=1920,H=1080,cams=8,sh=3,seed=1 --maxIteration 1500 --densifyStrategy 0 --warmupLength 100000 --progressTrain 0 --ssim 0.2"
run() { # tag env...
  tag=$1; shift
  env MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29900 + RANDOM % 50)) WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 "$@" divshot_amd/lib/gaussian_train $ARGS --outputPath /tmp/r6pc_$tag/it > /tmp/r6pc_$tag.log 2>&1
  printf "%-28s %s\n" "$tag" "$(grep -o '([0-9.]* it/s)' /tmp/r6pc_$tag.log | tail -1)  $(grep -c 'RCCL communicator up' /tmp/r6pc_$tag.log) comm"
}
for rep in 1 2; do
run plain DVS_FORCE_COMM=0
run rccl1_unchunked DVS_FORCE_COMM=1 DVS_A9_CHUNKS=1
run rccl1_chunks4 DVS_FORCE_COMM=1 DVS_A9_CHUNKS=4
run rccl1_chunks4_pipelined DVS_FORCE_COMM=1 DVS_A9_CHUNKS=4 DVS_EXCHANGE_PIPELINE=1
done
