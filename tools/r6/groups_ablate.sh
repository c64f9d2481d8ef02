#!/bin/bash
# round 6: where a detect_group wave's time goes -- diagnostic builds that stop it early (tools/variant.sh dab<n> -DRT_DETECT_ABLATE=<n>:
# 3 after the sort, 7 after the row means, 6 before the runs are gated, 4 no run statistics, 5 no hand-over), the reference's defaults,
# 4 096 streams, one lane, rocprofv3 kernel durations (timing only: the stopped builds return no or wrong records)
# usage (through gpurun): tools/r6/groups_ablate.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--steps 12 --warmup 3 --settle 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1 --group-detect on --sample-rate 300000 --streams 4096 --nperseg 256"
for v in analyze var_dab3 var_dab7 var_dab6 var_dab4 var_dab5; do
  d=$out/prof_$v
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common > $out/bench_$v.json 2> $out/bench_$v.err || { echo "failed $v"; tail -5 $out/bench_$v.err; exit 1; }
  echo "== $v"; grep "rt::detect_group\|rt::finalize" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-150
  rm -rf $d
done
