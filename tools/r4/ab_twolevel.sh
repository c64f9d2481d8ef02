#!/bin/bash
# nperseg 4096: a stream's earliest chunks half as long (default) against chunks of one length (RT_EXP_ONE_LEVEL=1, read by the DIAGNOSTIC library only: both variants run on librt_analyze_diag.so), same box
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_analyze_diag.so
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'frac', r['frac'], 'records', d['config']['records_per_step'], 'parity', d.get('parity',{}).get('streams_mismatched'))"; }
for rep in 1 2 3; do for v in two one; do
  unset RT_EXP_ONE_LEVEL; [ "$v" = one ] && export RT_EXP_ONE_LEVEL=1
  timeout -k 10 300 python3 bench.py --workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 20 --warmup 3 --settle 4 --isolated-steps 10 --parity-streams 8 2>>$out/err.txt | line "$v-level config5-share" >> $out/ab.txt
done; done
unset RT_EXP_ONE_LEVEL
for L in 48 56 87 112; do timeout -k 10 300 python3 bench.py --workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 10 --warmup 3 --settle 4 --isolated-steps 10 --parity-streams 0 --segs-per-chunk $L 2>>$out/err.txt | line "two-level L=$L" >> $out/ab.txt; done
sort $out/ab.txt
