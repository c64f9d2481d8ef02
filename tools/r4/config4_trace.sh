#!/bin/bash
# kernel stats of config 4's one-GPU-of-8 share (4 096 streams x 524 288 samples, nperseg 256) and of the 8 192-stream quarter: where the step's time
# beyond the scan goes (one lane: every kernel alone; two lanes: as the default runs it).   tools/r4/config4_trace.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for streams in 4096 8192; do
  for lanes in 1 2; do
    d=$out/s_${streams}_$lanes
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload config4 --total-streams $streams --lanes $lanes --steps 8 --warmup 2 --settle 3 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 > $out/bench_${streams}_lanes$lanes.json 2> $out/bench_${streams}_lanes$lanes.err || exit 1
    cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_${streams}_lanes$lanes.csv
    rm -rf $d
    echo "== $streams streams, $lanes lane(s)"; grep "rt::" $out/kernel_stats_${streams}_lanes$lanes.csv | cut -c1-120; python3 tools/show_bench.py $out/bench_${streams}_lanes$lanes.json | cut -c1-300
  done
done
