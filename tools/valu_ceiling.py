#!/usr/bin/env python3
"""The ceiling the composite kernels are actually against (SURVEY 8(d) "secondary ceilings"): vector-instruction ISSUE.
   frac = sum over instruction classes of (dynamic count per launch x issue cost in SIMD cycles) / (kernel cycles x SIMDs)
 * dynamic counts: rocprofv3 --pmc passes of the default bench command (tools/pmc.sh): SQ_INSTS_VALU and its classes ADD_F32 / MUL_F32 /
   FMA_F32 / TRANS_F32 / INT32 / CVT; "other" = SQ_INSTS_VALU - their sum (moves, compares, selects, min/max, DPP, permlane swaps, readlanes),
   priced with the kernel's STATIC mix of those (tools/valu_mix.py; not execution-weighted: stated);
 * issue costs: tools/ubench/valu_cost (ns per wave-instruction per SIMD at 8 waves per SIMD) x the clock that micro-benchmark ran at
   (GRBM_GUI_ACTIVE / duration of its own kernels, the same rocprofv3 pass) = SIMD cycles per wave-instruction;
 * kernel cycles: GRBM_GUI_ACTIVE of the kernel's launches (per XCD), SIMDs = 1024.
usage: tools/valu_ceiling.py pmc_a.txt pmc_b.txt ubench_costs.txt ubench_pmc.txt static_mix.json > profiles/rNN_valu_ceiling.json"""
import json
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic_json import kernel_source_sha

SIMDS = 1024


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s+(\S.*?)\s+((?:SQ|GRBM)_[A-Z0-9_]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
    return out


def durations(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s+(\S.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+\s+\d+\s+\d+\s+\d+", line)
        if m:
            out[m.group(1).strip()] = float(m.group(4))
    return out


def main():
    pa, pb, ub_txt, ub_pmc, mix_json = sys.argv[1:6]
    ctr = counters(pa)
    for k, v in counters(pb).items():
        for c, x in v.items():
            ctr.setdefault(k, {}).setdefault(c, x)
    dur = durations(pa)
    # the micro-benchmark: ns per wave-instruction per SIMD, and the clock it ran at
    ns = {}
    for line in open(ub_txt):
        m = re.match(r"(\S.*?)\s+wall\s+[0-9.]+ ms\s+->\s+([0-9.]+) ns per wave-instr", line)
        if m:
            ns[m.group(1).strip()] = float(m.group(2))
    uc, ud = counters(ub_pmc), durations(ub_pmc)
    clocks = []
    for k, v in uc.items():
        if "GRBM_GUI_ACTIVE" in v and k in ud and ud[k] > 50:
            g = v["GRBM_GUI_ACTIVE"]
            units = 8 if g / (ud[k] * 1e-6) > 3.0e9 else 1
            clocks.append(g / units / (ud[k] * 1e-6) / 1e9)
    ub_clock = sorted(clocks)[len(clocks) // 2] if clocks else 2.4
    cyc = {k: v * ub_clock for k, v in ns.items()}
    plain = (cyc.get("v_fma_f32", 0) + cyc.get("v_add_f32", 0) + cyc.get("v_mul_f32", 0)) / 3 or 3.4
    cost = {"fma_f32": cyc.get("v_fma_f32 3 distinct src", plain), "fmac_f32": cyc.get("v_fmac_f32", plain), "add_f32": cyc.get("v_add_f32", plain), "mul_f32": cyc.get("v_mul_f32", plain),
            "trans": (cyc.get("v_exp_f32", 0) + cyc.get("v_rcp_f32", 0) + cyc.get("v_log_f32", 0)) / 3 or 2.7 * plain,
            "int_other": (cyc.get("v_add_u32", plain) + cyc.get("v_lshlrev_b32", plain) + cyc.get("v_and_b32", plain) + cyc.get("v_mad_u32_u24", plain)) / 4,
            "cvt": cyc.get("v_cvt_f32_u32", plain), "mov": cyc.get("v_mov_b32", plain), "cmp": cyc.get("v_cmp_lt_f32 vcc", 1.5 * plain),
            "cndmask": cyc.get("v_cndmask_b32 sgpr mask", 1.5 * plain), "minmax": cyc.get("v_max_f32", 1.5 * plain),
            "dpp": cyc.get("v_mov_b32_dpp row_mirror", 1.5 * plain), "swap": cyc.get("v_permlane32_swap", 2.7 * plain), "lane_scalar": cyc.get("v_mov_b32", plain),
            "packed": cyc.get("v_pk_fma_f32", 1.6 * plain)}
    # (cndmask: the sgpr-mask form — the micro-benchmark's vcc form serialises on vcc, 9 ns, which is not what the kernels issue)
    mix = json.load(open(mix_json))["kernels"]
    out = {"_how": __doc__.strip().split("\nusage")[0], "kernel_source_sha16": kernel_source_sha(), "ubench_clock_GHz": ub_clock, "cycles_per_wave_instruction": cost, "kernels": {}}
    for want, mixkey in (("k_render_bwd_tr<true, false, 64>", "_Z15k_render_bwd_trILb1ELb0ELi64E"), ("k_render_fwd<false>", "_Z12k_render_fwdILb0E")):
        kc = next((v for k, v in ctr.items() if k.startswith(want)), None)
        kd = next((v for k, v in dur.items() if k.startswith(want)), None)
        km = next((v for k, v in mix.items() if k.startswith(mixkey)), None)
        if not kc or not kd or not km or "SQ_INSTS_VALU" not in kc:
            continue
        dyn = {"fma_f32": kc.get("SQ_INSTS_VALU_FMA_F32", 0), "add_f32": kc.get("SQ_INSTS_VALU_ADD_F32", 0), "mul_f32": kc.get("SQ_INSTS_VALU_MUL_F32", 0),
               "trans": kc.get("SQ_INSTS_VALU_TRANS_F32", 0), "int_other": kc.get("SQ_INSTS_VALU_INT32", 0), "cvt": kc.get("SQ_INSTS_VALU_CVT", 0)}
        # the FMA_F32 counter does not distinguish v_fmac (VOP2) from v_fma (VOP3, three register reads): split by the kernel's static ratio
        n_fmac, n_fma = km["by_class"].get("fmac_f32", 0), km["by_class"].get("fma_f32", 0)
        if n_fmac + n_fma:
            fm = dyn["fma_f32"]
            dyn["fmac_f32"] = fm * n_fmac / (n_fmac + n_fma)
            dyn["fma_f32"] = fm * n_fma / (n_fmac + n_fma)
        other = kc["SQ_INSTS_VALU"] - sum(dyn.values())
        oth_classes = ("mov", "cmp", "cndmask", "minmax", "dpp", "swap", "lane_scalar", "packed")
        tot_static = sum(km["by_class"].get(c, 0) for c in oth_classes) or 1
        other_cost = sum(km["by_class"].get(c, 0) * cost[c] for c in oth_classes) / tot_static
        issue_cycles = sum(dyn[c] * cost[c] for c in dyn) + max(other, 0.0) * other_cost
        g = kc.get("GRBM_GUI_ACTIVE")
        units = 8 if g and g / (kd * 1e-6) > 3.0e9 else 1
        kcycles = g / units if g else kd * 1e-6 * ub_clock * 1e9
        out["kernels"][want] = {"avg_us_under_pmc": kd, "kernel_cycles_per_launch": kcycles, "clock_GHz": kcycles / (kd * 1e-6) / 1e9,
                                "valu_wave_instructions_per_launch": kc["SQ_INSTS_VALU"], "by_counter_class": dyn, "other": other,
                                "other_cycles_per_instruction_from_static_mix": other_cost,
                                "static_share_of_other_in_kernel_text": tot_static / km["valu_total"], "dynamic_share_of_other": other / kc["SQ_INSTS_VALU"],
                                "issue_cycles_per_launch": issue_cycles, "frac_of_valu_issue_ceiling": issue_cycles / (SIMDS * kcycles),
                                "salu_per_launch": kc.get("SQ_INSTS_SALU"), "lds_per_launch": kc.get("SQ_INSTS_LDS")}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
