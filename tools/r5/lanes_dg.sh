#!/bin/bash
# the reference's defaults under a noise floor (4 096 streams): lanes 1 / 2 / 3 / 4 on one box
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'])"; }
common="--warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --steps 40 --sample-rate 300000 --streams 4096 --noise-dbw -88"
for rep in 1 2 3; do
for lanes in 1 2 3; do
  python3 bench.py $common --lanes $lanes 2>>$out/err.txt | line "defaults -88 dBW lanes $lanes" | tee -a $out/ab.txt
done
done
