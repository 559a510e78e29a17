#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) into the
per-kernel table committed under profiles/: calls, total/avg/min/max duration (us), share of GPU time, launch shape
and register use. Optionally PMC counter sums per kernel when the db came from a --pmc pass.
usage: tools/rocpd_summary.py results.db [--by-grid] > profiles/rNN_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "")


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv[2:]          # one row per (kernel, launch grid): separates the passes of a sort that share a kernel
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), "
        "max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) from kernels group by name" + (", grid_x" if by_grid else "") +
        " order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel-trace summary of {db}")
    print(f"# {'kernel':52s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'grid':>9s} {'wg':>5s} {'lds':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s}")
    for r in rows:
        print(f"  {short(r[0])[:52]:52s} {r[1]:6d} {r[2] / 1e3:12.1f} {r[3] / 1e3:10.2f} {r[4] / 1e3:10.2f} {r[5] / 1e3:10.2f} {100 * r[2] / total:6.2f} "
              f"{r[6]:9d} {r[7]:5d} {r[8]:6d} {r[9]:5d} {r[10]:5d} {r[11]:5d}")
    try:
        pmc = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                        "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters: kernel, counter, sum over dispatches, per-dispatch average")
        for name, ctr, v, cnt in pmc:
            print(f"  {short(name)[:52]:52s} {ctr:28s} {v:18.1f} {v / max(cnt, 1):16.1f}")


if __name__ == "__main__":
    main()
