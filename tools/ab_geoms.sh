#!/bin/bash
# A/B of the scan kernel over the large-nperseg geometries: tools/ab_geoms.sh <variant>...  ("default" = the in-tree library)
# per variant and geometry: load-only floor and one-lane scan time (tools/ablate_large.py)
for v in "$@"; do
  if [ "$v" = default ]; then unset RT_ANALYZE_LIB; else export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so; fi
  timeout -k 10 200 python tools/ablate_large.py 512 2048000 256
  timeout -k 10 200 python tools/ablate_large.py 1024 2400000 128
  timeout -k 10 200 python tools/ablate_large.py 2048 2048000 256
  timeout -k 10 200 python tools/ablate_large.py 4096 3200000 128
done
