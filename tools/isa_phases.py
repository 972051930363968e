"""Static instruction mix of a render kernel by PHASE of the bounce loop, from the compiler's assembly
with line information (.loc): every instruction is attributed to the source lines it was generated from,
and the lines to a phase -- set-up, the scene walks (the nearest-hit, container and shadow copies together:
the compiler interleaves their blocks), the rest of prepare_computations (normal, pattern), the light loop
(Phong, pow), return / combine, the tile store.  Instructions from the small helpers that every phase
uses (vector algebra, division, reciprocal) go to the phase of the code around them.

usage: python tools/isa_phases.py [kernel-substring]      (compiles raytracer_amd/csrc/rt_amd.hip with
-gline-tables-only into /tmp/rt_amd_lines.s first; about a minute)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_amd.hip")
OUT = "/tmp/rt_amd_lines.s"
K = sys.argv[1] if len(sys.argv) > 1 else "rt_kernelILb1ELb0ELb1ELb0ELb0ELb0E"

spec_m = __import__("importlib.util").util.spec_from_file_location("isa_mix", os.path.join(ROOT, "tools", "isa_mix.py"))


def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            if any(k in op for k in ("fma", "add_f64", "mul_f64", "fmac")):
                return "f64 add/mul/fma"
            if op.startswith("v_cmp"):
                return "compare"
            if "min" in op or "max" in op:
                return "f64 min/max"
            return "f64 other (rcp, rsq, div_*, ldexp, floor, cvt)"
        if op.startswith("v_cmp"):
            return "compare"
        if op.startswith("v_cndmask"):
            return "select"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "move"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "lane (scalar spills, broadcasts)"
        if "f32" in op:
            return "f32 arithmetic"
        return "integer / bit"
    if op.startswith(("s_load", "s_buffer")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("scratch", "buffer", "global", "flat")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


def source_marks():
    """Line numbers of the phase boundaries in the source, found by their comments."""
    lines = open(SRC).read().split("\n")

    def find(text, start=0):
        for i in range(start, len(lines)):
            if text in lines[i]:
                return i + 1
        raise SystemExit("marker not found: " + text)
    m = {}
    m["trace0"] = find("DI void trace(const DevScene& S, bool active")
    m["trace1"] = find("DI void renormalise(")
    m["kernel"] = find(") rt_kernel(DevScene S, DevJob J")
    m["prepare"] = find("---- World::prepare_computations", m["kernel"])
    m["light"] = find("---- per light: World::is_shadowed", m["kernel"])
    m["return"] = find("---- suspend, or return the colour", m["kernel"])
    m["store"] = find("---- write the tile", m["kernel"])
    m["end"] = find("// Canvas::to_ppm (src/window", m["kernel"])
    m["helpers_end"] = m["trace0"]  # everything above the walk: vector algebra, division, box tests, ballots
    m["normal0"] = m["trace1"]
    m["pattern0"] = find("__constant__ unsigned char PERLIN_PERM")
    m["schlick"] = find("DI double schlick(")
    m["combine"] = find("DI V3 combine(")
    m["primary"] = find("DI void primary_ray(")
    m["classk"] = find("__global__ void __launch_bounds__(64) tile_class_kernel")
    return m


def main():
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-Wno-unused-function", "-mllvm", "-disable-machine-licm", "-gline-tables-only", "-S", "--cuda-device-only", "-o", OUT, SRC]
        subprocess.check_call(cmd)
    m = source_marks()
    text = open(OUT).read()
    mm = re.search(r"^(_ZN\S*%s\S*):" % re.escape(K), text, flags=re.M)
    body = text[mm.end():text.index(".Lfunc_end", mm.end())].split("\n")
    file_of = {}
    for f in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', text, flags=re.M):
        file_of[int(f.group(1))] = f.group(3) or f.group(2)
    phase = "set-up"
    trace_no = 0
    in_trace = False
    names = ["walk: nearest hit", "walk: containers", "walk: shadow"]
    table = collections.defaultdict(collections.Counter)
    for l in body:
        t = l.strip()
        loc = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if loc:
            fname = file_of.get(int(loc.group(1)), "")
            line = int(loc.group(2))
            if not fname.endswith("rt_amd.hip"):
                continue  # the math library's headers (pow, cos, sqrt): stay in the phase that called them
            new = None
            if m["trace0"] <= line < m["trace1"]:
                # (the compiler lays the three inlined copies out in pieces, not one after the other: they are
                # counted together)
                new = "scene walks (3 inlined copies)"
                in_trace = True
                trace_no = max(trace_no, 1)
            elif line < m["helpers_end"]:
                new = phase  # helpers: where they are used
            else:
                in_trace_before = in_trace
                in_trace = False
                if m["normal0"] <= line < m["pattern0"]:
                    new = "prepare: normal_at"
                elif m["pattern0"] <= line < m["schlick"]:
                    new = "prepare: pattern"
                elif m["schlick"] <= line < m["combine"]:
                    new = "prepare: schlick"
                elif m["combine"] <= line < m["primary"]:
                    new = "return / combine"
                elif m["primary"] <= line < m["classk"]:
                    new = "set-up"
                elif m["kernel"] <= line < m["prepare"]:
                    new = "set-up" if trace_no == 0 else "prepare: rest"
                elif m["prepare"] <= line < m["light"]:
                    new = "prepare: rest"
                elif m["light"] <= line < m["return"]:
                    new = "light loop (Phong, pow)"
                elif m["return"] <= line < m["store"]:
                    new = "return / combine"
                elif m["store"] <= line < m["end"]:
                    new = "store, counters"
                else:
                    new = phase
            phase = new
            continue
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        table[phase][classify(t.split()[0])] += 1
    cols = ["f64 add/mul/fma", "f64 min/max", "f64 other (rcp, rsq, div_*, ldexp, floor, cvt)", "compare", "select", "move",
            "lane (scalar spills, broadcasts)", "f32 arithmetic", "integer / bit", "SALU", "SMEM", "VMEM", "LDS"]
    short = ["f64arith", "f64minmax", "f64other", "compare", "select", "move", "lane", "f32", "int/bit", "SALU", "SMEM", "VMEM", "LDS"]
    print("kernel", mm.group(1))
    print("%-32s %6s | " % ("phase", "VALU") + " ".join("%9s" % s for s in short))
    order = ["set-up", "scene walks (3 inlined copies)", "prepare: normal_at", "prepare: schlick", "prepare: pattern", "prepare: rest",
             "light loop (Phong, pow)", "return / combine", "store, counters"]
    tot = collections.Counter()
    for ph in order + [p for p in table if p not in order]:
        c = table.get(ph)
        if not c:
            continue
        valu = sum(c[k] for k in cols[:9])
        tot.update(c)
        print("%-32s %6d | " % (ph, valu) + " ".join("%9d" % c[k] for k in cols))
    print("%-32s %6d | " % ("total", sum(tot[k] for k in cols[:9])) + " ".join("%9d" % tot[k] for k in cols))


if __name__ == "__main__":
    main()
