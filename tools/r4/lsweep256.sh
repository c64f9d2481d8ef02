#!/bin/bash
# config 2 (nperseg 256, 8 000 segments per stream): whole path and scan time against segments per chunk, one and two lanes, same box.   tools/r4/lsweep256.sh <tag> [L...]
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms_alone', r.get('kernel_ms'), 'frac', r.get('frac'), 'scan_ms_concurrent', r.get('kernel_ms_concurrent'), 'records', d['config']['records_per_step'])"; }
for rep in 1 2; do
for L in ${@:-32 40 50 63 64}; do
  for lanes in 1 2; do
    timeout -k 10 300 python3 bench.py --lanes $lanes --segs-per-chunk $L --steps 100 --warmup 20 --isolated-steps 50 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "L=$L lanes $lanes rep $rep" >> $out/sweep.txt
  done
done
done
cat $out/sweep.txt
