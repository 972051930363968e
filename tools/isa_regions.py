"""Static instruction counts of a kernel split at its loop headers.  usage: isa_regions.py file.s [kernel-substring]"""
import collections, re, sys, importlib.util, os
spec = importlib.util.spec_from_file_location('isa_mix', os.path.join(os.path.dirname(__file__), 'isa_mix.py'))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
path = sys.argv[1]
K = sys.argv[2] if len(sys.argv) > 2 else "rt_kernelILb1ELb0ELb1ELb0ELb0ELb0E"
text = open(path).read()
mm = re.search(r"^(_ZN\S*%s\S*):" % re.escape(K), text, flags=re.M)
lines = text[mm.end():text.index(".Lfunc_end", mm.end())].split("\n")
marks = [i for i, l in enumerate(lines) if "Loop Header" in l] + [len(lines)]
prev = 0
tot = collections.Counter()
for i in marks:
    c = collections.Counter()
    for l in lines[prev:i]:
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;": continue
        c[m.classify(t.split()[0])] += 1
    tot += c
    print("%6d..%6d VALU %5d (f64a %4d minmax %3d f64o %3d cmp %3d cnd %3d mov %3d other %3d) salu %4d vmem %3d lds %3d" % (
        prev, i, sum(c[k] for k in m.VALU), c['f64_arith'], c['f64_minmax'], c['f64_other'], c['cmp'], c['cndmask'], c['mov'], c['valu_other'] + c['lane'], c['salu'], c['vmem'], c['lds']))
    prev = i
print("total VALU", sum(tot[k] for k in m.VALU))
