#!/bin/bash
# config 2 (the headline): lanes 2 / 3 / 4; the noisy defaults: lanes 3 / 5 / 6 -- one box, interleaved
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'])"; }
common="--warmup 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off"
for rep in 1 2 3; do
for lanes in 2 3 4; do
  python3 bench.py $common --steps 200 --warmup 30 --lanes $lanes 2>>$out/err.txt | line "config2 lanes $lanes" | tee -a $out/ab.txt
done
for lanes in 3 5 6; do
  python3 bench.py $common --settle 20 --steps 40 --sample-rate 300000 --streams 4096 --noise-dbw -88 --lanes $lanes 2>>$out/err.txt | line "defaults -88 dBW lanes $lanes" | tee -a $out/ab.txt
done
done
