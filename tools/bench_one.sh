#!/bin/bash
# usage (under gpurun, repo root): tools/bench_one.sh "<label>" <bench args...>  -- one line: kernel ms and Mrays/s
# (environment variables such as RT_AMD_LIB or RT_CSG_SHARE apply)
label=$1; shift
timeout -k 10 150 python bench.py --no-extras --steps ${STEPS:-40} --warmup 8 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s kernel %.3f ms  step %.3f ms  %.0f Mrays/s' % (sys.argv[1], d['kernel_ms'], d['ms_per_step'], d['value']))" "$label"
