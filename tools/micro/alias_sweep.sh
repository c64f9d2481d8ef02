#!/bin/bash
# Scan kernel at nperseg 256 with its loads aliased onto a footprint of 64 / 1024 / 8192 segments of stream 0
# (L2-resident / L2-sized / Infinity-Cache-resident) against the real buffer (HBM): which part of the kernel's
# time is the memory system.  Build the variants first:
#   tools/variant.sh base; tools/variant.sh a63 -DRT_EXP_ALIAS=63; tools/variant.sh a1023 -DRT_EXP_ALIAS=1023; tools/variant.sh a8191 -DRT_EXP_ALIAS=8191
# then on the GPU box:  bash tools/micro/alias_sweep.sh
for v in base a63 a1023 a8191; do
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so python tools/ablate_large.py 256 2048000 256 2>/dev/null | tail -1
done
