#!/bin/bash
# config 2 from the uint8 wire format: lanes 1 / 2 / 3, interleaved on one box
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'])"; }
common="--warmup 30 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --steps 200 --input u8"
for rep in 1 2 3; do
for lanes in 1 2 3; do
  python3 bench.py $common --lanes $lanes 2>>$out/err.txt | line "config2 uint8 lanes $lanes" | tee -a $out/ab.txt
done
done
