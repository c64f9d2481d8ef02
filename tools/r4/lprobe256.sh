#!/bin/bash
# why is the nperseg-256 scan alone faster with chunks of 20 than of 32?  config 2, one lane, scan alone (HIP events), L = 20 / 32, with and without candidates
# (a threshold no cell reaches: no emission, no step below a chunk, no sparse tail writes).   tools/r4/lprobe256.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: L', d['config'].get('segments_per_chunk'), 'scan_ms_alone', r.get('kernel_ms'), 'frac', r.get('frac'), 'ms/step', d['ms_per_step'], 'cells', d['config']['candidate_cells_per_step'], 'records', d['config']['records_per_step'])"; }
for rep in 1 2; do
for L in 32 20; do
  timeout -k 10 300 python3 bench.py --lanes 1 --segs-per-chunk $L --steps 60 --warmup 20 --isolated-steps 60 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "normal L=$L rep $rep" >> $out/probe.txt
  timeout -k 10 300 python3 bench.py --lanes 1 --segs-per-chunk $L --threshold-dbw 40 --steps 60 --warmup 20 --isolated-steps 60 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "no-candidates L=$L rep $rep" >> $out/probe.txt
  timeout -k 10 300 python3 bench.py --lanes 1 --segs-per-chunk $L --mode dense --steps 30 --warmup 10 --isolated-steps 30 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "dense L=$L rep $rep" >> $out/probe.txt
done
done
cat $out/probe.txt
