#!/bin/bash
# nperseg 512 at config-2 geometry, same box: library variants interleaved.   tools/r5/ab_n512.sh <tag> <variant>...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 30 --warmup 5 --settle 10 --isolated-steps 20 --cpu-streams 4 --parity-streams 4 --other-configs off --nperseg 512 --streams 512"
for rep in 1 2 3; do
  for v in "$@"; do
    lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
    RT_ANALYZE_LIB=$lib timeout -k 10 400 python3 bench.py $common 2>>$out/err.txt | line "$v" | tee -a $out/ab.txt
  done
done
