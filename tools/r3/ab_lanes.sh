#!/bin/bash
# persistent vs one-workgroup-per-item scans with one and two lanes on configs 3, 4 and the config-5 share: tools/r3/ab_lanes.sh <tag> <variant>...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # label lib args
  RT_ANALYZE_LIB=$2 timeout -k 10 300 python bench.py $3 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms_concurrent'], 'records', d['config']['records_per_step'])" >> $out/ab.txt || exit 1
}
for v in "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  for lanes in 1 2; do
  run "$v c3 lanes$lanes" $lib "--workload config3 --steps 12 --warmup 4 --settle 8 --lanes $lanes"
  run "$v c5 lanes$lanes" $lib "--workload config5 --total-streams 1024 --steps 12 --warmup 4 --settle 8 --lanes $lanes"
  run "$v c4 lanes$lanes" $lib "--workload config4 --total-streams 8192 --steps 12 --warmup 4 --settle 8 --lanes $lanes"
  done
done
sort -k2,3 -s $out/ab.txt
