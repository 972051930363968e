#!/bin/bash
# usage (under gpurun, repo root): tools/split_sweep.sh  -- small frames of every workload with the strip count the library picks
for a in "--workload csg --width 960 --height 540" "--workload csg --width 1920 --height 1080" "--workload glass --width 1920 --height 1080" "--workload glass --width 960 --height 540" "--width 1920 --height 1080" "--width 960 --height 540" "--workload field --width 1920 --height 1080" "--workload field --width 960 --height 540"; do bash tools/bench_one.sh "$a" --store f64 $a; done
