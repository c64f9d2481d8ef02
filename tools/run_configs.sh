#!/bin/bash
# The other BASELINE.json configurations at their full per-GPU sizes (parity-checked against the oracle on sampled
# streams by bench.py itself).  Not bench lines: results go to DESIGN.md section 5 and profiles/.
# usage: tools/run_configs.sh [out-file] [extra bench flags, e.g. --lanes 1]
out=${1:-/dev/stdout}; shift
{
timeout -k 10 900 python bench.py --workload config3 --steps 5 --warmup 2 --settle 3 --isolated-steps 5 --cpu-streams 16 "$@" 2>/dev/null | tail -1
timeout -k 10 900 python bench.py --workload config4 --steps 5 --warmup 2 --settle 3 --isolated-steps 5 --cpu-streams 64 "$@" 2>/dev/null | tail -1
timeout -k 10 900 python bench.py --workload config5 --total-streams 1024 --steps 5 --warmup 2 --settle 3 --isolated-steps 5 --cpu-streams 16 "$@" 2>/dev/null | tail -1
} > $out
