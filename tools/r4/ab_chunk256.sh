#!/bin/bash
# config 2: the handle's own chunk length (25) against 32, interleaved on one box, scan alone (one lane) and whole path (two lanes).   tools/r4/ab_chunk256.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: L', d['config'].get('segments_per_chunk'), 'value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms_alone', r.get('kernel_ms'), 'frac', r.get('frac'), 'scan_ms_concurrent', r.get('kernel_ms_concurrent'))"; }
for rep in 1 2 3; do
  for L in 0 32 20; do
    timeout -k 10 300 python3 bench.py --segs-per-chunk $L --steps 100 --warmup 20 --isolated-steps 50 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "config2 two lanes segs-per-chunk=$L rep $rep" >> $out/ab.txt
  done
done
cat $out/ab.txt
