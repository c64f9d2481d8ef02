#!/bin/bash
# round 6: what finalize_records' 120 us per 4 096 streams are -- timing-only builds (tools/variant.sh dab8 -DRT_DETECT_ABLATE=8: no per-stream
# words to pinned host memory; dab9: the records to device memory instead of the pinned pool), reference defaults, one lane, rocprofv3
# usage (through gpurun): tools/r6/finalize_ablate.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--steps 12 --warmup 3 --settle 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 256"
for v in analyze var_dab8 var_dab9; do
  d=$out/prof_$v
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common > $out/bench_$v.json 2> $out/bench_$v.err || { echo "failed $v"; tail -5 $out/bench_$v.err; }
  echo "== $v"; grep "rt::detect_group\|rt::finalize" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-150
  rm -rf $d
done
