#!/bin/bash
# BASELINE configs 3, 4 (all streams), 5 (share), same box: library variants interleaved.   tools/r5/ab_configs.sh <tag> <variant>...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'records', d['config']['records_per_step'])"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 4 --no-cpu-baseline --parity-streams 0 --other-configs off"
for rep in 1 2; do
for cfg in "config3" "config4" "config5 --total-streams 1024" "config2"; do
  for v in "$@"; do
    lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
    RT_ANALYZE_LIB=$lib timeout -k 10 400 python3 bench.py $common --workload $cfg 2>>$out/err.txt | line "$v ${cfg%% *}" | tee -a $out/ab.txt
  done
done
done
