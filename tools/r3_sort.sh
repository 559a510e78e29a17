#!/bin/bash
cd "$(dirname "$0")/.."
timeout 180 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort_pairs or edge_cases or C1_10k or G2_2k" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --profile-iters 5 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
st=d['pipeline']['stages']
print('views/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'single-view stages: depth_sort', round(st['depth_sort']['ms'],4), 'tile_sort', round(st['tile_sort']['ms'],4), 'strict', round(d['strict_single_view']['ms_per_view'],4))"
