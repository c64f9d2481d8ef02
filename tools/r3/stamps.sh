#!/bin/bash
# per-stage cycle sums of the scan step (tools/variant.sh stamps -DRT_STAMPS): tools/r3/stamps.sh <tag>
out=gpurun_out/${1:-r3i}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "4096 3200000 512" "2048 2048000 512" "1024 2400000 1024" "256 2048000 256"; do
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_stamps.so timeout -k 10 200 python tools/ablate_large.py $g > $out/run.txt 2> $out/run.err || exit 1
  tail -1 $out/run.txt >> $out/stamps.txt
  grep RT_STAMPS $out/run.err | tail -1 >> $out/stamps.txt
done
cat $out/stamps.txt
