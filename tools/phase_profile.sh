#!/bin/bash
# usage (under gpurun, repo root): tools/phase_profile.sh <bench args...>  -- phase shares of a wavefront's time
# from the -DRT_PROFILE build (variants/prof.so): s_memtime stamps around the phases of the bounce loop.
export RT_AMD_LIB=$PWD/variants/prof.so
timeout -k 10 150 python bench.py --no-extras --steps 3 --warmup 1 "$@" 2>&1 >/dev/null | grep RT_PROFILE | tail -1 | python -c "
import sys, re
l = sys.stdin.read()
d = {k: float(v) for k, v in re.findall(r'(\w+)=(\d+)', l)}
w = d['wave_cycles']
closest, shadow = d['closest_cycles'], d['shadow_cycles']
prep = d['prepare_cycles'] - closest
light = d['lighting_cycles'] - shadow
ret = d['return_cycles']
rest = w - closest - shadow - prep - light - ret
print('waves %d rounds/wave %.2f walks %d records/walk %.1f | closest %.1f%% shadow %.1f%% prepare %.1f%% light %.1f%% return %.1f%% setup+store %.1f%% | mean wave %.0f ticks' % (
    d['waves'], d['rounds'] / d['waves'], d['traces'], d['visits'] / max(d['traces'], 1), 100 * closest / w, 100 * shadow / w, 100 * prep / w, 100 * light / w, 100 * ret / w, 100 * rest / w, w / d['waves']))"
