#!/bin/bash
# round 6: the nperseg-128 defaults line five times on one box (three lanes), then the bare command once more (host sinks)
tag=${1:-r6l}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'])"; }
common="--steps 20 --warmup 5 --settle 10 --isolated-steps 5 --no-cpu-baseline --parity-streams 0 --other-configs off --sample-rate 300000 --streams 4096"
for rep in 1 2 3; do
  for n in 128 256; do
    for lanes in 3 1; do
      timeout -k 10 300 python3 bench.py $common --lanes $lanes --nperseg $n 2>>$out/err.txt | line "defaults nperseg $n lanes $lanes" | tee -a $out/ab.txt
    done
  done
done
( time timeout -k 10 600 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err ); echo "bench rc=$?"
python3 tools/show_bench.py $out/bench_n1.json | cut -c1-300
python3 -c "
import json; d=json.loads(open('$out/bench_n1.json').read().strip().splitlines()[-1]); h=d['host_sinks']; print({k:v for k,v in h.items() if k!='note' and k!='timed_region_ends_at'})"
