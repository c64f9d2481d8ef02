#!/bin/bash
# The other BASELINE.json configurations at their full per-GPU sizes (parity-checked against the oracle on sampled
# streams by bench.py itself).  Not bench lines: results go to DESIGN.md section 5 and profiles/.
# usage: tools/run_configs.sh [out-file]
out=${1:-/dev/stdout}
{
timeout 900 python bench.py --streams 4096 --sample-rate 2400000 --nperseg 1024 --window hann --steps 5 --warmup 2 --cpu-streams 16 2>/dev/null | tail -1
timeout 900 python bench.py --streams 32768 --sample-rate 2048000 --seconds 0.256 --steps 5 --warmup 2 --cpu-streams 64 2>/dev/null | tail -1
timeout 900 python bench.py --streams 1024 --sample-rate 3200000 --nperseg 4096 --trains --steps 5 --warmup 2 --cpu-streams 16 2>/dev/null | tail -1
} > $out
