#!/bin/bash
# one-lane scan time of library variants on configs 2 / 3 / 5: tools/diag/ab_variants.sh <variant>...
for cfg in "config2" "config3 --total-streams 1024" "config5 --total-streams 512"; do
  for v in "$@"; do
    lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
    RT_ANALYZE_LIB=$lib timeout -k 10 300 python bench.py --workload $cfg --lanes 1 --no-cpu-baseline --steps 40 --warmup 10 --settle 20 --isolated-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $v', 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'value', d['value'], 'records', d['config']['records_per_step'])" || echo "$cfg $v FAILED"
  done
done
