#!/bin/bash
# Build an A/B variant of librt_analyze.so with extra -D switches: tools/variant.sh <name> [flags...]
# (always with -DRT_DIAG: the product build refuses laboratory switches and ignores their environment variables, csrc/rt_diag.h)
# -> pyradiotracking_amd/librt_var_<name>.so ; run it with RT_ANALYZE_LIB=<path> python bench.py ...
set -e
name=$1; shift
cd "$(dirname "$0")/../pyradiotracking_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -Wno-pass-failed \
      -DRT_DIAG "$@" -I../../include -o ../librt_var_$name.so rt_analyze.hip rt_match.cpp rt_format.cpp
