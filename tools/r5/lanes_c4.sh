#!/bin/bash
# config-4 geometry at per-rank batch sizes: lanes 2 / 3
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'])"; }
common="--warmup 3 --settle 4 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --steps 12 --workload config4"
for rep in 1 2; do
for n in 1024 4096 8192; do
for lanes in 2 3; do
  python3 bench.py $common --total-streams $n --lanes $lanes 2>>$out/err.txt | line "config4 $n streams lanes $lanes" | tee -a $out/ab.txt
done
done
done
