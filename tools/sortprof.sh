export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sorttest -o p -- python $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 --profile-iters 0 > $R/gpurun_out/sorttest_bench.json 2>/dev/null
cd $R && python tools/rocpd_summary.py gpurun_out/prof_sorttest/p_results.db 2>&1 | grep -E "k_sort|k_tile" 
python -c "
import json; d=json.loads([l for l in open('gpurun_out/sorttest_bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
