#!/bin/bash
out=gpurun_out/${1:-r3n}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "256 2048000 256" "4096 3200000 512"; do
  set -- $g
  RT_STAMPS_DUMP=$PWD/$out/dump_$1.bin RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_stamps.so timeout -k 10 200 python tools/ablate_large.py $g > $out/run.txt 2> $out/run.err || exit 1
  python tools/r3/timeline.py $out/dump_$1.bin > $out/timeline_$1.txt
  cat $out/timeline_$1.txt
done
