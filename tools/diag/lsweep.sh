#!/bin/bash
# chunk-length sweep, one lane, isolated scan time
for cfg in "config3 --total-streams 1024" "config5 --total-streams 512" "config2"; do
  for L in 0 24 48 64 96; do
    timeout -k 10 300 python bench.py --workload $cfg --lanes 1 --no-cpu-baseline --steps 8 --warmup 3 --settle 6 --isolated-steps 0 --segs-per-chunk $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg L=$L', 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'value', d['value'])"
  done
done
