#!/bin/bash
# stage ablation of the scan kernel (variants built by tools/ablate.sh): tools/diag/ablate_run.sh <nperseg> <fs> <streams>
for n in ${ABL:-1 2 3 5 6 8 7}; do
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_ablate_$n.so timeout -k 10 200 python tools/ablate_large.py $1 $2 $3 2>/dev/null | grep "^N=" | sed "s/^/stage $n: /"
done
timeout -k 10 200 python tools/ablate_large.py $1 $2 $3 2>/dev/null | grep "^N=" | sed "s/^/full:    /"
