#!/bin/bash
# does the four-lane cliff come from the runtime's four hardware queues?  lanes 3 / 4 / 6 with GPU_MAX_HW_QUEUES at its default (4) and at 8
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'])"; }
common="--isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off"
for rep in 1 2; do
for q in 4 8; do
export GPU_MAX_HW_QUEUES=$q
for lanes in 3 4 6; do
  python3 bench.py $common --steps 200 --warmup 30 --lanes $lanes 2>>$out/err.txt | line "config2 queues $q lanes $lanes" | tee -a $out/ab.txt
  python3 bench.py $common --warmup 5 --settle 20 --steps 40 --sample-rate 300000 --streams 4096 --noise-dbw -88 --lanes $lanes 2>>$out/err.txt | line "defaults -88 dBW queues $q lanes $lanes" | tee -a $out/ab.txt
done
done
done
