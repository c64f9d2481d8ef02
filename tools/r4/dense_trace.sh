#!/bin/bash
# kernel trace of the dense path (RT_MODE_DENSE) at config 2, clean input and a floor 2 dB over the threshold, one and two lanes
tag=${1:-r4dense}
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lanes in 1 2; do for noise in "" "--noise-dbw -88"; do
  n=$(echo "$noise" | tr -d ' -'); d=$out/s_${lanes}_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --mode dense --lanes $lanes $noise --steps 30 --warmup 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 > $out/bench_${lanes}_$n.json 2> $out/err_${lanes}_$n.txt || exit 1
  cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_lanes${lanes}_$n.csv; rm -rf $d
  echo "lanes $lanes $noise: $(tail -1 $out/bench_${lanes}_$n.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; grep "rt::" $out/kernel_stats_lanes${lanes}_$n.csv | cut -d, -f1-4 | cut -c1-110
done; done
