#!/bin/bash
# alternating A/B of two library variants: tools/diag/ab_alt.sh "<bench args>" <variantA> <variantB> [rounds]
args=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq $n); do
  for v in $a $b; do
    lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --isolated-steps 0 $args 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'scan_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'value', d['value'])" || echo "$v FAILED"
  done
done
