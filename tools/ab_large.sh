#!/bin/bash
# scan-kernel timing of variants at the larger nperseg geometries: tools/ab_large.sh <variant>...
for v in "$@"; do
  for g in "512 2048000 256" "1024 2400000 128" "2048 2048000 256" "4096 3200000 128"; do
    RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so timeout -k 10 200 python tools/ablate_large.py $g 2>/dev/null | tail -1
  done
done
