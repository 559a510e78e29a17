#!/usr/bin/env python3
"""Static instruction mix of the composite kernels (CPU; hipcc cross-compiles): which VALU instructions k_render_fwd and k_render_bwd_tr
consist of, by issue-cost class. The dynamic per-class COUNTS come from SQ counters (tools/valu_ceiling.py); the counters know
ADD_F32 / MUL_F32 / FMA_F32 / TRANS_F32 / INT32 / CVT, and everything else a kernel issues (moves, compares, selects, min/max, DPP and
permlane operations, readlanes) lands in "other = SQ_INSTS_VALU - the sum of those". This tool says what that remainder is made of,
statically (per instruction of the kernel's text, not weighted by execution), so that the remainder can be priced with the measured
issue costs of tools/ubench/valu_cost.hip.
usage: tools/valu_mix.py > profiles/rNN_valu_static_mix.json"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics -fno-slp-vectorize".split()
KERNELS = {"render.hip": ["_Z12k_render_fwdILb0E"], "render_tr.hip": ["_Z15k_render_bwd_trILb1ELb0ELi64E", "_Z15k_render_bwd_trILb1ELb1ELi64E"]}

# cost classes of tools/ubench/valu_cost.hip (ns per wave-instruction per SIMD): plain 1.4 | cmp / min / max / cndmask / DPP 2.05 |
# transcendental and v_permlane*_swap 3.75 | packed 2.3
def klass(op, operands):
    if "dpp" in op or "dpp" in operands or "row_" in operands or "quad_perm" in operands or "wave_" in operands:
        return "dpp"
    if op.startswith(("v_permlane",)):
        return "swap"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "cmp"
    if op.startswith(("v_max", "v_min", "v_med3")):
        return "minmax"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane_scalar"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "mov"
    if op.startswith(("v_fmac", "v_mac_f")):
        return "fmac_f32"            # VOP2: two register reads
    if op.startswith(("v_fma", "v_mad_f")):
        return "fma_f32"             # VOP3: three register reads (slower: measured)
    if op.startswith(("v_add_f", "v_sub_f", "v_subrev_f")):
        return "add_f32"
    if op.startswith("v_mul_f"):
        return "mul_f32"
    if op.startswith("v_cvt"):
        return "cvt"
    return "int_other"          # integer add / shift / logic / mad_u32 / bfe / bitop ...


def main():
    out = {"_how": "static VALU instruction mix of the shipped composite kernels (instructions of the kernel text, NOT execution-weighted); classes "
                   "by issue cost (tools/ubench/valu_cost.hip)", "kernels": {}}
    with tempfile.TemporaryDirectory() as d:
        for src, prefixes in KERNELS.items():
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(ROOT, "divshot_amd", "csrc", src), "-o", os.path.join(d, "o.o"), "--save-temps=obj"],
                                  stderr=subprocess.DEVNULL, cwd=os.path.join(ROOT, "divshot_amd", "csrc"))
            asm = open(os.path.join(d, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
            for pre in prefixes:
                m = re.search(r"^(" + re.escape(pre) + r"\w*):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M)
                if not m:
                    continue
                counts, total = {}, 0
                for line in m.group(2).splitlines():
                    mm = re.match(r"\s+(v_[a-z0-9_]+)\s*(.*)", line)
                    if not mm:
                        continue
                    k = klass(mm.group(1), mm.group(2))
                    counts[k] = counts.get(k, 0) + 1
                    total += 1
                salu = len(re.findall(r"^\s+s_(?!waitcnt|nop|endpgm|barrier|branch|cbranch)", m.group(2), re.M))
                lds = len(re.findall(r"^\s+ds_", m.group(2), re.M))
                out["kernels"][m.group(1)[:48]] = {"valu_total": total, "by_class": dict(sorted(counts.items())), "salu": salu, "lds": lds}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
