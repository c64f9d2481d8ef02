#!/bin/bash
# lane-centric candidate emission in the generic scan (default) against the commit before (variant prev: sixteen ballots per emitting step), interleaved on one box.
#   tools/r4/ab_emission.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms_alone', r.get('kernel_ms'), 'frac', r.get('frac'), 'scan_ms_concurrent', r.get('kernel_ms_concurrent'), 'records', d['config']['records_per_step'])"; }
run() { name=$1; shift; for rep in 1 2; do for v in analyze var_prev; do
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --parity-streams 0 "$@" 2>>$out/err.txt | line "$name $v rep $rep" >> $out/ab.txt; done; done; }
run config2_two_lanes --steps 100 --warmup 20 --isolated-steps 60
run config2_one_lane --lanes 1 --steps 100 --warmup 20 --isolated-steps 0
run default_geometry_clean --sample-rate 300000 --streams 4096 --steps 30 --warmup 5 --settle 20 --isolated-steps 20
run default_geometry_floor88 --sample-rate 300000 --streams 4096 --steps 30 --warmup 5 --settle 20 --isolated-steps 0 --noise-dbw -88 --mode runfilter
run config4_eighth --workload config4 --total-streams 4096 --steps 16 --warmup 4 --settle 3 --isolated-steps 10
run config3 --workload config3 --steps 8 --warmup 2 --settle 3 --isolated-steps 5
run config2_floor90 --noise-dbw -90 --threshold-dbw -90 --steps 60 --warmup 10 --isolated-steps 0
cat $out/ab.txt
