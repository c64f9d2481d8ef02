#!/bin/bash
# nperseg 4096, config-5 share, one lane: scan time against segments per chunk (and two lanes at the default):
#   tools/r4/lsweep4096.sh <tag> <L>...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--workload config5 --total-streams 1024 --no-cpu-baseline --steps 10 --warmup 3 --settle 4 --isolated-steps 10 --parity-streams 0"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'frac', r['frac'], 'detect_ms', r['detect_kernel_ms'], 'records', d['config']['records_per_step'])"; }
for L in "$@"; do
  timeout -k 10 300 python3 bench.py $common --lanes 1 --segs-per-chunk $L 2>>$out/err.txt | line "L=$L lanes 1" >> $out/lsweep.txt || echo "L=$L FAILED" >> $out/lsweep.txt
done
timeout -k 10 300 python3 bench.py $common --lanes 2 2>>$out/err.txt | line "default L lanes 2" >> $out/lsweep.txt
timeout -k 10 300 python3 bench.py $common --lanes 1 2>>$out/err.txt | line "default L lanes 1" >> $out/lsweep.txt
cat $out/lsweep.txt
