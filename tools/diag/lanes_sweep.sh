#!/bin/bash
for cfg in "config4 --total-streams 8192" "config5 --total-streams 512" "config3 --total-streams 1024" "config2"; do
  for lanes in 1 2 3 4; do
    timeout -k 10 300 python bench.py --workload $cfg --lanes $lanes --no-cpu-baseline --steps 12 --warmup 3 --settle 6 --isolated-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg lanes=$lanes', 'value', d['value'], 'ms/step', d['ms_per_step'], 'whole', d['roofline']['whole_path_frac'])"
  done
done
