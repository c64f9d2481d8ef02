#!/bin/bash
# round 6: what would a cheaper sparse detection buy?  Whole path with diagnostic builds whose detect_bucket waves stop early
# (tools/variant.sh dab3 -DRT_DETECT_ABLATE=3: after the sort; dab6: before the runs are gated; both return NO records: timing only)
# against the product, same box: the reference's defaults clean and under a noise floor (three lanes), config 4 with all streams (one lane)
# usage (through gpurun): tools/r6/detect_upper_bound.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'])"; }
common="--steps 30 --warmup 5 --settle 10 --isolated-steps 10 --no-cpu-baseline --parity-streams 0 --other-configs off"
for rep in 1 2; do
for v in analyze var_dab6 var_dab3; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so
  timeout -k 10 300 python3 bench.py $common --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256 2>>$out/err.txt | line "defaults clean $v" | tee -a $out/bench.txt || exit 1
  timeout -k 10 300 python3 bench.py $common --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256 --noise-dbw -88 2>>$out/err.txt | line "defaults floor -88 $v" | tee -a $out/bench.txt || exit 1
  [ $rep = 1 ] && { timeout -k 10 300 python3 bench.py $common --steps 10 --workload config4 --lanes 1 2>>$out/err.txt | line "config4 all streams $v" | tee -a $out/bench.txt || exit 1; }
done
done
