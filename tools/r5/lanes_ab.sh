#!/bin/bash
# lanes 1 .. 4 on config 4 (all streams), config 2 and the reference defaults (clean): tools/r5/lanes_ab.sh <tag>
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'records', d['config']['records_per_step'])"; }
common="--warmup 3 --settle 4 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off"
for rep in 1 2; do
for lanes in 1 2 3 4 6; do
  python3 bench.py $common --steps 10 --workload config4 --lanes $lanes 2>>$out/err.txt | line "config4 lanes $lanes" | tee -a $out/ab.txt
done
for lanes in 1 2 3 4; do
  python3 bench.py $common --steps 40 --lanes $lanes 2>>$out/err.txt | line "config2 lanes $lanes" | tee -a $out/ab.txt
  python3 bench.py $common --steps 20 --sample-rate 300000 --streams 4096 --lanes $lanes 2>>$out/err.txt | line "defaults clean lanes $lanes" | tee -a $out/ab.txt
done
done
