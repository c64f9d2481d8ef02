#!/bin/bash
# this round's library against last round's final one (librt_var_r03.so, built from commit 6a2cea9), same box, whole path:
#   config 2 complex64 and uint8, one and two lanes.   tools/r4/ab_round.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'records', d['config']['records_per_step'])"; }
for rep in 1 2; do for v in default r03; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  for lanes in 2 1; do
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --lanes $lanes --no-cpu-baseline --steps 100 --warmup 10 --parity-streams 0 2>>$out/err.txt | line "$v c64 lanes $lanes" >> $out/ab.txt
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --lanes $lanes --input u8 --no-cpu-baseline --steps 100 --warmup 10 --parity-streams 0 2>>$out/err.txt | line "$v u8 lanes $lanes" >> $out/ab.txt
  done
done; done
sort $out/ab.txt
