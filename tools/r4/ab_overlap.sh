#!/bin/bash
# detection on its own lower-priority stream (default build) against everything in order on one stream (variant prev = the commit
# before), same box: config-5 share, config 3 and config 2 (one and two lanes), whole path.   tools/r4/ab_overlap.sh <tag> <variant>
tag=$1; var=$2
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'detect_ms', r['detect_kernel_ms'], 'records', d['config']['records_per_step'], 'parity', d.get('parity',{}).get('streams_mismatched'))"; }
for v in default $var default $var; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 20 --warmup 3 --settle 4 --isolated-steps 10 --parity-streams 8 2>>$out/err.txt | line "$v config5-share lanes 1" >> $out/ab.txt
  RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --workload config3 --lanes 1 --no-cpu-baseline --steps 10 --warmup 2 --settle 3 --isolated-steps 5 --parity-streams 8 2>>$out/err.txt | line "$v config3 lanes 1" >> $out/ab.txt
  for lanes in 1 2; do
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --lanes $lanes --no-cpu-baseline --steps 100 --warmup 10 --parity-streams 8 2>>$out/err.txt | line "$v config2 lanes $lanes" >> $out/ab.txt
  done
done
cat $out/ab.txt
