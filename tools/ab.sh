#!/bin/bash
# A/B of kernel variants on one box: tools/ab.sh "<bench args>" <variant>...   (variants built by tools/variant.sh)
# prints value / ms_per_step / kernel_ms per variant, one-lane and two-lane
args=$1; shift
for v in "$@"; do
  for lanes in 1 2; do
    lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --isolated-steps 0 --steps 100 --warmup 20 --lanes $lanes $args 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v lanes $lanes: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'records', d['config']['records_per_step'])" || echo "$v lanes $lanes FAILED"
  done
done
