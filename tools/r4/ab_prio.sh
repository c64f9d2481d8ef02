#!/bin/bash
# experiment: priority of the detection's stream against the scan's (RT_EXP_DETECT_PRIO = low (default) | high | same | off), same box
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'records', d['config']['records_per_step'])"; }
for rep in 1 2; do for pr in low high same off; do
  export RT_EXP_DETECT_PRIO=$pr
  timeout -k 10 300 python3 bench.py --workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 20 --warmup 3 --settle 4 --isolated-steps 10 --parity-streams 0 2>>$out/err.txt | line "$pr config5-share lanes 1" >> $out/ab.txt
  timeout -k 10 300 python3 bench.py --workload config3 --lanes 1 --no-cpu-baseline --steps 10 --warmup 2 --settle 3 --isolated-steps 5 --parity-streams 0 2>>$out/err.txt | line "$pr config3 lanes 1" >> $out/ab.txt
  timeout -k 10 300 python3 bench.py --lanes 2 --no-cpu-baseline --steps 100 --warmup 10 --parity-streams 0 2>>$out/err.txt | line "$pr config2 lanes 2" >> $out/ab.txt
  timeout -k 10 300 python3 bench.py --lanes 1 --no-cpu-baseline --steps 100 --warmup 10 --parity-streams 0 2>>$out/err.txt | line "$pr config2 lanes 1" >> $out/ab.txt
done; done
sort $out/ab.txt
