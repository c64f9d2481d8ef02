#!/bin/bash
# round 6 (the round-5 review's item 8): nperseg 1024 with an LDS-DMA landing zone at two workgroups per CU (tools/variant.sh dma1024 -DRT_EXP_DMA1024=1)
# against the product (three workgroups per CU, register prefetch), BASELINE config 3 on one box, interleaved
tag=${1:-r6j}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--workload config3 --steps 10 --warmup 3 --settle 4 --isolated-steps 6 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1"
for rep in 1 2 3; do
  for v in product dma1024; do
    if [ $v = product ]; then unset RT_ANALYZE_LIB; else export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so; fi
    timeout -k 10 400 python3 bench.py $common 2>>$out/err.txt | line "config3 $v" | tee -a $out/ab.txt
  done
done
