#!/bin/bash
# A/B of library variants on the dense path (config 2 and the reference's default geometry, two lanes), interleaved on one box:
#   tools/r3/ab_dense.sh <tag> <variant>...      (variant "default" = the product build)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # label lib args
  RT_ANALYZE_LIB=$2 timeout -k 10 300 python bench.py $3 --no-cpu-baseline --isolated-steps 0 2>>$out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', 'value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms_concurrent'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'records', d['config']['records_per_step'], 'parity', (d.get('parity') or {}).get('streams_mismatched'))" >> $out/ab.txt || exit 1
}
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  run "$v c2-dense rep$rep" $lib "--mode dense --steps 40 --warmup 10"
  run "$v c2-dense-noisy rep$rep" $lib "--mode dense --noise-dbw -88 --threshold-dbw -90 --steps 40 --warmup 10"
  run "$v default-geometry-dense rep$rep" $lib "--mode dense --sample-rate 300000 --streams 4096 --noise-dbw -88 --steps 20 --warmup 5"
  run "$v c3-dense rep$rep" $lib "--mode dense --workload config3 --total-streams 1024 --steps 10 --warmup 3 --lanes 1"
done
done
sort -k2,2 -s $out/ab.txt
