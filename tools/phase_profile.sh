#!/bin/bash
# usage (under gpurun, repo root): tools/phase_profile.sh <bench args...>
# Phase shares of a wavefront's time and the walk's own work (walks, records visited) from the
# -DRT_PROFILE build (variants/prof.so): s_memtime stamps (shader clocks) around the phases of the bounce loop.
# Prints one readable line; with PHASE_JSON=<file> also appends one JSON line to that file.
export RT_AMD_LIB=$PWD/variants/prof.so
timeout -k 10 150 python bench.py --no-extras --steps 3 --warmup 1 "$@" 2>&1 >/dev/null | grep RT_PROFILE | tail -1 | python -c "
import sys, re, json, os
l = sys.stdin.read()
d = {k: float(v) for k, v in re.findall(r'(\w+)=(\d+)', l)}
w = d['wave_cycles']
closest, shadow = d['closest_cycles'], d['shadow_cycles']
prep = d['prepare_cycles'] - closest
light = d['lighting_cycles'] - shadow
ret = d['return_cycles']
rest = w - closest - shadow - prep - light - ret
out = {'args': ' '.join(sys.argv[1:]), 'waves': d['waves'], 'rounds': d['rounds'], 'walks': d['traces'], 'records_visited': d['visits'],
       'records_per_walk': d['visits'] / max(d['traces'], 1), 'mean_wave_clocks': w / d['waves'],
       'share': {'nearest_hit_walks': closest / w, 'shadow_walks': shadow / w, 'prepare': prep / w, 'lighting': light / w,
                 'return': ret / w, 'setup_store': rest / w},
       'share_of_walks': {'leaf_records': d.get('leaf_cycles', 0) / (closest + shadow), 'csg_end_records': d.get('csg_end_cycles', 0) / (closest + shadow)}}
print('waves %d rounds/wave %.2f walks %d records/walk %.1f | closest %.1f%% shadow %.1f%% prepare %.1f%% light %.1f%% return %.1f%% setup+store %.1f%% | of the walks: leaf records %.1f%% csg end records %.1f%% | mean wave %.0f clocks' % (
    d['waves'], d['rounds'] / d['waves'], d['traces'], out['records_per_walk'], 100 * closest / w, 100 * shadow / w, 100 * prep / w, 100 * light / w, 100 * ret / w, 100 * rest / w,
    100 * out['share_of_walks']['leaf_records'], 100 * out['share_of_walks']['csg_end_records'], w / d['waves']))
if os.environ.get('PHASE_JSON'):
    open(os.environ['PHASE_JSON'], 'a').write(json.dumps(out) + '\n')
" "$@"
