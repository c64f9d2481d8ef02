#!/bin/bash
# A/B of library variants on the default bench (two lanes and one lane), one box: tools/r3/ab_default.sh <tag> <variant>...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  for lanes in 2 1; do
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --isolated-steps 0 --steps 100 --warmup 20 --lanes $lanes 2>>$out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v lanes $lanes rep $rep: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms_concurrent'], 'detect_ms', d['roofline']['detect_kernel_ms'])" >> $out/ab.txt || exit 1
  done
done
done
cat $out/ab.txt
