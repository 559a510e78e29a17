#!/usr/bin/env python3
"""Build profiles/rNN_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc.sh + rocpd_summary.py).
usage: tools/make_traffic_json.py pmc_fetch.txt pmc_write.txt [steps_of_the_profiled_command] > profiles/rNN_traffic.json
With the step count of the profiled bench command (tools/pmc.sh: 3 timed + 1 warm-up = 4) every kernel also gets launches_per_step
(= calls / steps; kernels launched fewer times than there were steps are set-up work) and the file a counter_bytes_per_step total."""
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """sha256 over the kernel sources the counters were collected on (divshot_amd/csrc: *.hip and *.h — the device code and its headers — sorted by name) — bench.py
    recomputes it and says whether the committed counters still belong to the code it is timing."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "divshot_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "divshot_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\s+(\S.*?)\s+" + counter + r"\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m:
            out[m.group(1).strip()] = float(m.group(3))
    return out


def calls(path):
    """kernel -> launches, from the kernel-trace table of the same summary file"""
    out = {}
    for line in open(path):
        m = re.match(r"\s+(\S.*?)\s+(\d+)\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+\d+\s+\d+\s+\d+", line)
        if m:
            out[m.group(1).strip()] = int(m.group(2))
    return out


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ncalls = calls(sys.argv[1])
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
                      "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
        if steps and ncalls.get(k, 0) >= steps:
            kernels[k]["launches_per_step"] = ncalls[k] // steps      # (the command's single extra forward adds a launch or two: dropped)
    how = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh) on bench.py's default workload, C3, "
           "1x MI355X. Units are KB (x1024). Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE "
           "reports half the bytes of wide coalesced reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. Calibrated on "
           "k_preprocess_fwd (236 B read per splat). The factor 2 is NOT calibrated for the narrow gathers of the composite kernels "
           "(their figure is an upper bound), the counters include Infinity-Cache hits, and cross-XCD fp32 atomics are counted as writes.")
    out = {"_how": how, "workload": "C3", "views_per_launch": 8, "kernel_source_sha16": kernel_source_sha(), "kernels": kernels}
    if steps:
        out["steps_of_profiled_command"] = steps
        out["counter_bytes_per_step"] = int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches_per_step"] for v in kernels.values()
                                                if "launches_per_step" in v))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
