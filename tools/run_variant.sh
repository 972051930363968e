#!/bin/bash
# usage (under gpurun, repo root): tools/run_variant.sh <variant> <bench args...>  -- one bench run with variants/<variant>.so in place of the library
cp raytracer_amd/csrc/librt_amd.so /tmp/orig.so
cp variants/$1.so raytracer_amd/csrc/librt_amd.so; shift
timeout -k 10 120 python bench.py --no-extras --store f64 "$@"
rc=$?
cp /tmp/orig.so raytracer_amd/csrc/librt_amd.so
exit $rc
