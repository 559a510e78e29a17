#!/bin/bash
# libgstrain's one-view-per-step iteration (the per-rank shape of an 8-GPU run) at C3 size: plain, and with the exchange over a 1-rank RCCL
# communicator with 1 / 4 A9 chunks. it/s from the host's progress lines (at step 500 and 1000).
cd "$(dirname "$0")/../divshot_amd/lib"
run() { echo "== $1"; shift; env "$@" ./gaussian_train --inputPath synthetic:N=1000000,W=1920,H=1080,cams=8,sh=3,seed=1 --maxIteration 1001 --densifyStrategy 0 --warmupLength 100000 --progressTrain 0 --outputPath /tmp/plg/it 2>/dev/null | grep "it/s" | tail -1; }
run "plain (no communicator)" DVS_X=0
run "RCCL 1 rank, A9 unchunked" DVS_FORCE_COMM=1 DVS_A9_CHUNKS=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29711
run "RCCL 1 rank, 4 chunks" DVS_FORCE_COMM=1 DVS_A9_CHUNKS=4 MASTER_ADDR=127.0.0.1 MASTER_PORT=29712
run "RCCL 1 rank, 8 chunks" DVS_FORCE_COMM=1 DVS_A9_CHUNKS=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=29713
run "RCCL 1 rank, plain all-reduce of all rows" DVS_FORCE_COMM=1 DVS_EXCHANGE=allreduce MASTER_ADDR=127.0.0.1 MASTER_PORT=29714
