#!/bin/bash
# usage (repo root, no GPU needed): tools/asan_cpu_suite.sh [pytest args]
# Builds the host library and the oracle with AddressSanitizer + UndefinedBehaviorSanitizer and runs the
# CPU test suite against them.  The Python interpreter itself is not instrumented, so the sanitizer
# runtime is preloaded; leak checking is off (the interpreter and the HIP runtime keep memory until exit).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
make -C $ROOT/raytracer_amd/host asan
make -C $ROOT/oracle asan
export RT_HOST_LIB=$ROOT/raytracer_amd/host/librt_host_asan.so
export RT_ORACLE_LIB=$ROOT/oracle/liboracle_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cd $ROOT
exec python -m pytest tests -x -q -m "not gpu" "$@"
