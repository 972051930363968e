#!/bin/bash
# usage (under gpurun, repo root): tools/phase_set.sh <out.jsonl> <workload>:<W>x<H> ...   (f64 store, like tools/profile_set.sh)
out=$1; shift
mkdir -p $(dirname $out); : > $out
for cfg in "$@"; do
  w=${cfg%%:*}; size=${cfg##*:}; W=${size%%x*}; H=${size##*x}
  echo "== $cfg"
  PHASE_JSON=$out bash tools/phase_profile.sh --store f64 --workload $w --width $W --height $H
done
