#!/bin/bash
# default geometry (4 096 x 300 kS/s, nperseg 256), kernel traces: clean sparse / clean runfilter / floor -88 runfilter, lanes 1 and 2
#   tools/r5/trace_dg.sh <tag> [lib]
tag=${1:-r5dg}; lib=${2:-$PWD/pyradiotracking_amd/librt_analyze.so}
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--sample-rate 300000 --streams 4096 --steps 20 --warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off"
run() { # name, flags...
  name=$1; shift
  d=$out/s_$name
  RT_ANALYZE_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common "$@" > $out/bench_$name.json 2> $out/bench_$name.err || { echo "FAILED $name"; tail -5 $out/bench_$name.err; return 1; }
  cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_$name.csv
  python3 tools/r5/steps.py $(ls $d/*/*kernel_trace.csv | head -1) > $out/steps_$name.txt 2>&1
  rm -rf $d
  python3 -c "
import json,sys
d=json.loads([l for l in open('$out/bench_$name.json') if l.startswith('{')][-1]); print('$name', d['value'], d['ms_per_step'], d['config']['mode'])"
}
run clean_sparse_l1 --lanes 1 --mode sparse &&
run clean_runfilter_l1 --lanes 1 --mode runfilter &&
run noisy_runfilter_l1 --lanes 1 --mode runfilter --noise-dbw -88 &&
run noisy_runfilter_l2 --lanes 2 --mode runfilter --noise-dbw -88 &&
run clean_sparse_l2 --lanes 2 --mode sparse
# plain (no profiler)
for l in 1 2; do
RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py $common --lanes $l --mode runfilter --noise-dbw -88 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain noisy runfilter lanes $l', d['value'], d['ms_per_step'])"
done
