#!/bin/bash
# the round's evidence for the exact pre-filter at the reference's defaults: PMC traffic (two passes each, floor at / 2 dB over the threshold)
# and one- / two-lane kernel-trace csv of the AUTO run with the floor at -88 dBW.   tools/r5/dg_profiles.sh <tag>
tag=$1
bash tools/profile_runfilter_traffic.sh $tag -151.8 > /dev/null
bash tools/profile_runfilter_traffic.sh $tag -153.8 > /dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$PWD/gpurun_out/prof_$tag
common="--sample-rate 300000 --streams 4096 --steps 20 --warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --noise-dbw -88"
for lanes in 1 2; do
  d=$out/s_$lanes
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common --mode auto --lanes $lanes > $out/bench_auto_lanes$lanes.json 2> $out/bench_auto_lanes$lanes.err || exit 1
  cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_auto_lanes$lanes.csv
  python3 tools/r5/steps.py $(ls $d/*/*kernel_trace.csv | head -1) > $out/steps_auto_lanes$lanes.txt 2>&1
  rm -rf $d
done
ls $out
