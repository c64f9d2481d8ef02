#!/bin/bash
# A/B of library variants on configs 2 (one and two lanes), 3, 4 (8 192 streams) and the config-5 share, interleaved on one box:
#   tools/r3/ab_configs.sh <tag> <variant>...      (variant "default" = the product build)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # label lib args
  RT_ANALYZE_LIB=$2 timeout -k 10 300 python bench.py $3 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms_concurrent'], 'frac', d['roofline']['frac_concurrent'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'records', d['config']['records_per_step'])" >> $out/ab.txt || exit 1
}
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  run "$v c2 lanes2 rep$rep" $lib "--steps 100 --warmup 20 --lanes 2"
  run "$v c2 lanes1 rep$rep" $lib "--steps 100 --warmup 20 --lanes 1"
  run "$v c3 lanes1 rep$rep" $lib "--workload config3 --steps 12 --warmup 4 --settle 8 --lanes 1"
  run "$v c4 lanes2 rep$rep" $lib "--workload config4 --total-streams 8192 --steps 12 --warmup 4 --settle 8 --lanes 2"
  run "$v c5 lanes1 rep$rep" $lib "--workload config5 --total-streams 1024 --steps 12 --warmup 4 --settle 8 --lanes 1"
done
done
sort -k2,3 -s $out/ab.txt
