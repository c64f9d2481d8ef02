#!/bin/bash
# is finalize_records bound by its writes to pinned host memory?  Timing-only variant devrec (-DRT_EXP_DEV_RECORDS: records to device memory) against the product,
# kernel stats at config 4's eighth and quarter and the default geometry, one lane.   tools/r4/finalize_probe.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in analyze var_devrec; do
  for w in "--workload config4 --total-streams 4096" "--workload config4 --total-streams 8192" "--sample-rate 300000 --streams 4096"; do
    n=$(echo $w | tr -d ' -' | cut -c1-30); d=$out/s_${v}_$n
    RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $w --lanes 1 --steps 8 --warmup 2 --settle 3 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 > $out/bench_${v}_$n.json 2> $out/bench_${v}_$n.err
    echo "== $v $w"; grep "rt::final" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-110; python3 tools/show_bench.py $out/bench_${v}_$n.json | cut -c1-120
    rm -rf $d
  done
done
