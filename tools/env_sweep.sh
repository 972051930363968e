#!/bin/bash
# usage (under gpurun, repo root): tools/env_sweep.sh VAR "v1 v2 ..." "<bench args>" ["<bench args>" ...]  -- kernel ms per value of an environment knob
VAR=$1; VALS=$2; shift; shift
for v in $VALS; do export $VAR=$v; for a in "$@"; do bash tools/bench_one.sh "$VAR=$v $a" --store f64 $a; done; done
