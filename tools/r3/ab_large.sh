#!/bin/bash
# A/B of library variants on the persistent-grid scans only (config 3 and the config-5 share, one lane), interleaved on one box:
#   tools/r3/ab_large.sh <tag> <variant>...      (variant "default" = the product build)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # label lib args
  RT_ANALYZE_LIB=$2 timeout -k 10 300 python bench.py $3 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms_concurrent'], 'frac', d['roofline']['frac_concurrent'])" >> $out/ab.txt || exit 1
}
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  run "$v c3 lanes1 rep$rep" $lib "--workload config3 --steps 12 --warmup 4 --settle 8 --lanes 1"
  run "$v c5 lanes1 rep$rep" $lib "--workload config5 --total-streams 1024 --steps 12 --warmup 4 --settle 8 --lanes 1"
done
done
sort -k2,3 -s $out/ab.txt
