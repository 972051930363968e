#!/bin/bash
# usage (under gpurun, repo root): tools/mix_run.sh  -- tiles per cost class, records visited by kind, rounds and the
# longest wavefront's rounds for a few configurations, from the -DRT_PROFILE_MIX build of the library (build it first:
# the Makefile's flags plus -DRT_PROFILE_MIX, output variants/mix.so).
export RT_AMD_LIB=$PWD/variants/mix.so
for a in "--workload glass --width 1920 --height 1080" "--workload field --width 1920 --height 1080" "--workload csg --width 1920 --height 1080" "--width 1920 --height 1080" "--workload glass --width 3840 --height 2160"; do
  echo "== $a"
  timeout -k 10 150 python bench.py --no-extras --steps 2 --warmup 1 --store f64 $a 2>&1 >/dev/null | grep RT_PROFILE_MIX | tail -3
done
