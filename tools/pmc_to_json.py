"""Turns the per-configuration PMC summaries (pmc_<workload>_<W>x<H>.txt, written by
tools/profile_set.sh) and kernel stats into profiles/pmc.json, which bench.py reads for
`roofline.traffic`, the executed-work figures and the issue fractions.

usage: python tools/pmc_to_json.py profiles/r3            (a directory of pmc_*.txt / kernel_stats_*.csv)

Units and corrections
  * FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE tallies 128-byte read requests
    as 64 bytes, so fetched bytes = 2 x FETCH_SIZE x 1024.
  * SQ_INSTS_VALU_*_F64 are wavefront instructions: executed flops = 64 lanes x (ADD + MUL + 2 FMA + TRANS),
    an upper bound (the average instruction has SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU active lanes).
  * SQ_ACTIVE_INST_VALU, SQ_INST_CYCLES_SMEM, SQ_WAVE_CYCLES, SQ_WAIT_* are in quad-cycles (4 clocks), summed
    over the SIMDs; GRBM_GUI_ACTIVE is in clocks, summed over the 8 XCDs.  Calibration
    (profiles/r3/issue_calib.txt, tools/issue_calib.sh): for kernels that are one unbroken stream of a single
    vector instruction on every SIMD,
        4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    reads 0.97-0.99 for every instruction that takes 4 clocks or more per wavefront (f64 add / mul / fma /
    min / max / compare, 32-bit compare / select / min / max / bfi / convert: 1 quad each; v_rcp_f32 2 quads,
    v_rcp_f64 / v_rsq_f64 4 quads) and 1.9 for the 32-bit instructions that take 2 clocks (v_mov_b32,
    v_and_b32, v_sub_f32, v_fma_f32): the counter charges at least one quad per instruction.  So
        valu_issue_frac      = 4 x SQ_ACTIVE_INST_VALU / (1024 x clocks of the launch)          (an upper bound)
        valu_issue_frac_min  = the same with 2 clocks instead of 4 for every instruction that MAY be a 2-clock
                               one: all but the classes known to take 4 or more (the F64 classes, CVT, INT32 --
                               which holds the carry arithmetic, not v_mov / v_and --, INT64, TRANS_F32)  (a lower bound)
    and the truth lies between the two (compares, selects, min / max are unclassified too and take 4).
  * SQ_INST_CYCLES_SMEM turned out to equal SQ_INSTS_SMEM (one quad per instruction, not its latency) on the
    micro-benchmark that is nothing but a chain of dependent record fetches (tools/ubench/smem_latency), so
    the scalar-fetch chain is priced from the walk's own counts instead (records visited per wavefront, from the
    -DRT_PROFILE build, times the measured latency of a dependent fetch): see bench.py.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024  # 256 CUs x 4
N_XCD = 8


def derive(c, e):
    """Fractions that follow from the counters alone (no wall clock): see the module docstring."""
    gui = c.get("GRBM_GUI_ACTIVE")
    act = c.get("SQ_ACTIVE_INST_VALU")
    if gui and act:
        clocks = gui / N_XCD  # clocks of the launch
        e["launch_clocks"] = clocks
        e["valu_busy_quads"] = act
        e["valu_issue_frac"] = 4.0 * act / (N_SIMD * clocks)
        slow = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                           "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32",
                                           "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_TRANS_F32"))
        fast = c.get("SQ_INSTS_VALU", 0.0) - slow
        if "SQ_INSTS_VALU_INT32" in c and "SQ_INSTS_VALU_ADD_F64" in c:
            e["valu_insts_in_2clock_classes"] = fast
            e["valu_issue_frac_min"] = (4.0 * act - 2.0 * fast) / (N_SIMD * clocks)
        if c.get("SQ_INSTS_SALU"):
            # a SIMD gets one issue turn every four clocks and a turn takes at most one scalar instruction
            e["salu_issue_frac"] = 4.0 * c["SQ_INSTS_SALU"] / (N_SIMD * clocks)
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        if c.get("SQ_WAIT_ANY"):
            e["wait_any_share"] = c["SQ_WAIT_ANY"] / wc
        if c.get("SQ_WAIT_INST_ANY"):
            e["wait_issue_share"] = c["SQ_WAIT_INST_ANY"] / wc
        if gui:
            e["mean_resident_waves_per_simd"] = 4.0 * wc / (N_SIMD * gui / N_XCD)
    if c.get("SQ_INSTS_VALU"):
        f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                          "SQ_INSTS_VALU_TRANS_F64"))
        if f64:
            e["f64_share_of_valu_insts"] = f64 / c["SQ_INSTS_VALU"]
    if c.get("SQC_DCACHE_REQ"):
        e["scalar_cache_requests"] = c["SQC_DCACHE_REQ"]
        e["scalar_cache_misses"] = c.get("SQC_DCACHE_MISSES")


def main():
    src = sys.argv[1]
    path = os.path.join(ROOT, "profiles", "pmc.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*.txt"))):
        key = re.match(r"pmc_(.*)\.txt", os.path.basename(f)).group(1)
        c = {}
        for line in open(f):
            m = re.match(r"(\S+)\s+n=(\d+) mean=(\S+)", line)
            if m:
                c[m.group(1)] = float(m.group(3))
        if "FETCH_SIZE" not in c:
            continue
        e = {
            "fetch_size_kb": c["FETCH_SIZE"],
            "write_size_kb": c["WRITE_SIZE"],
            "bytes_per_launch": int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
            "valu_insts": c.get("SQ_INSTS_VALU"),
            "salu_insts": c.get("SQ_INSTS_SALU"),
            "smem_insts": c.get("SQ_INSTS_SMEM"),
            "vmem_insts": c.get("SQ_INSTS_VMEM"),
            "f64_add": c.get("SQ_INSTS_VALU_ADD_F64"),
            "f64_mul": c.get("SQ_INSTS_VALU_MUL_F64"),
            "f64_fma": c.get("SQ_INSTS_VALU_FMA_F64"),
            "f64_trans": c.get("SQ_INSTS_VALU_TRANS_F64"),
            "waves": c.get("SQ_WAVES"),
            "source": os.path.relpath(f, ROOT),
        }
        if e["f64_add"] is not None:
            e["executed_flops"] = 64.0 * (e["f64_add"] + e["f64_mul"] + 2.0 * e["f64_fma"] + (e["f64_trans"] or 0.0))
        if c.get("SQ_ACTIVE_INST_VALU"):
            e["mean_active_lanes"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] if c.get("SQ_THREAD_CYCLES_VALU") else None
        derive(c, e)
        ks = os.path.join(src, "kernel_stats_%s.csv" % key)
        if os.path.exists(ks):
            for r in csv.DictReader(open(ks)):
                if "rt_kernel" in r["Name"]:
                    e["rocprof_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
                    e["rocprof_kernel_calls"] = int(r["Calls"])
                    break
        if e.get("launch_clocks") and e.get("rocprof_kernel_avg_ms"):
            e["clock_ghz"] = e["launch_clocks"] / (e["rocprof_kernel_avg_ms"] * 1e6)
        out[key] = e
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path, "with", len(out), "configurations")


if __name__ == "__main__":
    main()
