#!/bin/bash
# detect_bucket<false> at config 4's 4 096-stream share, one lane, diagnostic builds (RT_DETECT_ABLATE: 1 no row means, 2 no sort, 3 stop after the sort)
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-analyze var_dab1 var_dab3 var_dab6 var_dab4 var_dab5}; do
  d=$out/s_$v
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload config4 --total-streams 4096 --lanes 1 --steps 8 --warmup 2 --settle 3 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 > $out/bench_$v.json 2> $out/bench_$v.err
  echo "== $v"; grep "rt::detect\|rt::final" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-110
  rm -rf $d
done
