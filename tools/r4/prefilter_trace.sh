#!/bin/bash
# kernel trace of the exact run-length pre-filter (RT_MODE_RUNFILTER / AUTO) at the reference's default geometry
# (4 096 streams x 300 kS/s, nperseg 256, floor -88 dBW against the -90 dBW threshold), one lane and two lanes:
# which kernels the step's time goes to, and the gaps between them.   tools/r4/prefilter_trace.sh <tag>
tag=${1:-r4pf}
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--sample-rate 300000 --streams 4096 --steps 20 --warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --noise-dbw -88"
for lanes in 1 2; do
  for mode in runfilter auto; do
    d=$out/s_${mode}_$lanes
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common --mode $mode --lanes $lanes > $out/bench_${mode}_lanes$lanes.json 2> $out/bench_${mode}_lanes$lanes.err || exit 1
    cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_${mode}_lanes$lanes.csv
    cp $(ls $d/*/*kernel_trace.csv | head -1) $out/kernel_trace_${mode}_lanes$lanes.csv
    rm -rf $d
    python3 tools/timeline.py $out/kernel_trace_${mode}_lanes$lanes.csv > $out/timeline_${mode}_lanes$lanes.txt 2>&1 || true
    rm -f $out/kernel_trace_${mode}_lanes$lanes.csv  # (tens of MiB: only the timeline travels back)
  done
done
# the same without the profiler, for the plain figure
for lanes in 1 2; do
  timeout -k 10 300 python3 bench.py $common --mode runfilter --lanes $lanes | tail -1 > $out/plain_runfilter_lanes$lanes.json || exit 1
done
cut -c1-150 $out/kernel_stats_runfilter_lanes1.csv
