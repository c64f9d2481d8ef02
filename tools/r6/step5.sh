#!/bin/bash
# round 6: stft_scan64 with quarter 2 of the next segment requested a step ahead (RT_W64_Q2AHEAD=1, variant) against the product, same box;
# then the N = 1 points of the sharded curves (config 5 with all 8 192 streams on one GPU) for profiles/n1_reference.json
tag=${1:-r6i}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 20 --warmup 3 --settle 4 --isolated-steps 10 --no-cpu-baseline --parity-streams 4 --other-configs off --lanes 1 --workload config5 --total-streams 1024"
for rep in 1 2 3; do
  for v in product w64_q2ahead; do
    if [ $v = product ]; then unset RT_ANALYZE_LIB; else export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so; fi
    timeout -k 10 300 python3 bench.py $common 2>>$out/err.txt | line "config5 share $v" | tee -a $out/ab.txt
  done
done
unset RT_ANALYZE_LIB
timeout -k 10 600 python3 bench.py --workload config5 --steps 10 --warmup 2 --settle 3 --isolated-steps 3 --no-cpu-baseline --parity-streams 4 --other-configs off 2>>$out/err.txt | tee $out/config5_all_streams_n1.json | line "config5 all 8192 streams on one GPU"
