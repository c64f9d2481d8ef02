#!/bin/bash
# nperseg 256: scan time against segments per chunk at config 2 and at config 4's eighth (a wave's four lane groups read four chunks L x 2 KiB apart:
# powers of two collide in the memory system), same box.   tools/r4/lsweep256b.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms_alone', r.get('kernel_ms'), 'frac', r.get('frac'), 'scan_ms_concurrent', r.get('kernel_ms_concurrent'), 'records', d['config']['records_per_step'])"; }
for L in 32 19 20 21 23 25 27 29 31 33 35 32; do
  for lanes in 1 2; do
    timeout -k 10 300 python3 bench.py --lanes $lanes --segs-per-chunk $L --steps 100 --warmup 20 --isolated-steps 50 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "config2 L=$L lanes $lanes" >> $out/sweep.txt
  done
done
for L in 32 21 22 23 25 26 27 29 31 32; do
  for lanes in 1 2; do
    timeout -k 10 300 python3 bench.py --workload config4 --total-streams 4096 --lanes $lanes --segs-per-chunk $L --steps 16 --warmup 4 --settle 3 --isolated-steps 10 --no-cpu-baseline --parity-streams 0 2>>$out/err.txt | line "config4_eighth L=$L lanes $lanes" >> $out/sweep.txt
  done
done
cat $out/sweep.txt
