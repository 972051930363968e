"""Static instruction mix of the scene-walk loop of the main render kernel, from the compiler's
assembly (raytracer_amd/csrc/rt_amd.s, `make -C raytracer_amd/csrc asm`).

usage: python tools/isa_mix.py [kernel-mangled-name-substring]

Prints, for the first inner loop of the kernel (the nearest-hit walk, trace<MODE_CLOSEST>), one line
per basic block with its instructions by class, and the totals.  Which blocks a record kind runs
through is read off the branch structure (see profiles/r2/isa_mix_trace_loop.txt for the annotated
result of one build)."""
import collections
import re
import sys

KERNEL = sys.argv[1] if len(sys.argv) > 1 else "rt_kernelILb1ELb0ELb1ELb0ELb0ELb0E"


def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            if any(k in op for k in ("fma", "add_f64", "mul_f64", "fmac")):
                return "f64_arith"
            if "min" in op or "max" in op:
                return "f64_minmax"
            if op.startswith("v_cmp"):
                return "cmp"
            return "f64_other"   # rcp, rsq, div_scale/fmas/fixup, ldexp, floor, cvt
        if op.startswith("v_cmp"):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "mov"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "lane"
        return "valu_other"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("scratch", "buffer", "global", "flat")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


VALU = ("f64_arith", "f64_minmax", "f64_other", "cmp", "cndmask", "mov", "lane", "valu_other")


def main():
    text = open("raytracer_amd/csrc/rt_amd.s").read()
    m = re.search(r"^(_ZN\S*%s\S*):" % re.escape(KERNEL), text, flags=re.M)
    start = m.end()
    end = text.index(".Lfunc_end", start)
    lines = text[start:end].split("\n")
    # first inner loop: from its header label to the next loop header at the same depth
    heads = [i for i, l in enumerate(lines) if "Loop Header: Depth=2" in l]  # the walks (they may contain small loops)
    lo, hi = heads[0] - 1, heads[1] - 1
    print("kernel %s\nloop: lines %d..%d of the function" % (m.group(1)[:90], lo, hi))
    block, rows, total = "entry", [], collections.Counter()
    cur = collections.Counter()
    for l in lines[lo:hi]:
        lab = re.match(r"(\.LBB\d+_\d+):", l) or re.match(r"; %bb\.(\d+):", l)
        if lab:
            if sum(cur.values()):
                rows.append((block, cur))
            block, cur = lab.group(1), collections.Counter()
            continue
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        c = classify(t.split()[0])
        cur[c] += 1
        total[c] += 1
    if sum(cur.values()):
        rows.append((block, cur))
    cols = list(VALU) + ["salu", "smem", "vmem", "lds"]
    print("%-12s %5s | " % ("block", "VALU") + " ".join("%10s" % c for c in cols))
    for name, c in rows:
        print("%-12s %5d | " % (name, sum(c[k] for k in VALU)) + " ".join("%10d" % c[k] for k in cols))
    print("%-12s %5d | " % ("TOTAL", sum(total[k] for k in VALU)) + " ".join("%10d" % total[k] for k in cols))


if __name__ == "__main__":
    main()
