#!/bin/bash
# config 2, same box: this round's library with the detection on its own stream (default) / in order (the DIAGNOSTIC library, RT_EXP_STREAMS=1: the product reads no environment) against last round's (r03)
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'conc_ms', r.get('kernel_ms_concurrent'), 'detect_ms', r['detect_kernel_ms'])"; }
for rep in 1 2 3; do for v in default onestream r03; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  unset RT_EXP_STREAMS; [ "$v" = onestream ] && { export RT_EXP_STREAMS=1; lib=$PWD/pyradiotracking_amd/librt_analyze_diag.so; }
  for lanes in 2 1; do
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --lanes $lanes --no-cpu-baseline --steps 200 --warmup 30 --parity-streams 0 2>>$out/err.txt | line "$v c64 lanes $lanes" >> $out/ab.txt
  done
  RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py --lanes 2 --input u8 --no-cpu-baseline --steps 200 --warmup 30 --parity-streams 0 2>>$out/err.txt | line "$v u8 lanes 2" >> $out/ab.txt
done; done
sort $out/ab.txt
