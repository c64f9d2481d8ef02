#!/bin/bash
# round 6: per-kernel durations (rocprofv3 --kernel-trace --stats, one lane) of the sparse detection with and without detect_group:
# the reference's defaults (4 096 streams) and config 4's eighth (4 096 streams x 524 288 samples)
# usage (through gpurun): tools/r6/groups_trace.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--steps 12 --warmup 3 --settle 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1"
for g in off auto; do
  for wl in "defaults --sample-rate 300000 --streams 4096 --nperseg 256" "config4_eighth --workload config4 --total-streams 4096"; do
    name=${wl%% *}; flags=${wl#* }
    d=$out/prof_${name}_$g
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common --group-detect $g $flags > $out/bench_${name}_$g.json 2> $out/bench_${name}_$g.err || { echo "failed $name $g"; tail -5 $out/bench_${name}_$g.err; exit 1; }
    echo "== $name groups $g"; grep "rt::" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-150
    cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_${name}_groups_$g.csv; rm -rf $d
  done
done
