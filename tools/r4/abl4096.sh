#!/bin/bash
# nperseg 4096, config-5 share, one lane: scan time of diagnostic variants of the one-wave-per-segment kernel
# (tools/variant.sh w64abl<n> -DRT_W64_ABL=n ...) next to the product build and the micro-benchmark, same box.
#   tools/r4/abl4096.sh <tag> <L> <variant>...
tag=$1; L=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 10 --warmup 3 --settle 4 --isolated-steps 10 --parity-streams 0 --segs-per-chunk $L"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'frac', r['frac'], 'detect_ms', r['detect_kernel_ms'], 'records', d['config']['records_per_step'])"; }
for v in "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py $common 2>>$out/err.txt | line "$v L=$L" >> $out/abl.txt || echo "$v FAILED" >> $out/abl.txt
done
cat $out/abl.txt
