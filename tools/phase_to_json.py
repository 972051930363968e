"""Turns the JSON lines tools/phase_set.sh wrote (one per configuration, from the -DRT_PROFILE build) into
profiles/walks.json, keyed <workload>_<W>x<H>, which bench.py reads for the walk's own work (walks, records
visited, phase shares of a wavefront's time).   usage: python tools/phase_to_json.py profiles/r3/phases.jsonl"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "walks.json")
out = json.load(open(path)) if os.path.exists(path) else {}
for line in open(sys.argv[1]):
    e = json.loads(line)
    a = e["args"]
    w = re.search(r"--workload (\S+)", a)
    key = "%s_%sx%s" % (w.group(1) if w else "default", re.search(r"--width (\d+)", a).group(1), re.search(r"--height (\d+)", a).group(1))
    e["source"] = os.path.relpath(sys.argv[1], ROOT)
    out[key] = e
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path, "with", len(out), "configurations")
