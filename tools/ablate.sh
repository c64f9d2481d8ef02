#!/bin/bash
# Build diagnostic variants of librt_analyze.so that stop the scan step after stage n
# (RT_ABLATE=n) next to the product build; bench them with RT_ANALYZE_LIB=... python bench.py
set -e
cd "$(dirname "$0")/../pyradiotracking_amd/csrc"
for n in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -Wno-pass-failed \
        -DRT_ABLATE=$n -I../../include -o ../librt_ablate_$n.so rt_analyze.hip
done
