"""Turns the per-configuration PMC summaries (pmc_<workload>_<W>x<H>.txt, written by
tools/profile_set.sh) and kernel stats into profiles/pmc.json, which bench.py reads for
`roofline.traffic` and the executed-work figures.

usage: python tools/pmc_to_json.py profiles/r2            (a directory of pmc_*.txt / kernel_stats_*.csv)

Units and corrections: FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE tallies
128-byte read requests as 64 bytes, so fetched bytes = 2 x FETCH_SIZE x 1024.  The SQ_INSTS_VALU_*_F64
counters are wavefront instructions: executed flops = 64 lanes x (ADD + MUL + 2 FMA + TRANS), an upper
bound (the average instruction has SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU active lanes).
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        ks = os.path.join(src, "kernel_stats_%s.csv" % key)
        if os.path.exists(ks):
            for r in csv.DictReader(open(ks)):
                if "rt_kernel" in r["Name"]:
                    e["rocprof_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
                    e["rocprof_kernel_calls"] = int(r["Calls"])
                    break
        out[key] = e
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path, "with", len(out), "configurations")


if __name__ == "__main__":
    main()
