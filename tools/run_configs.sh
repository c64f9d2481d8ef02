#!/bin/bash
# The other BASELINE.json configurations at their full per-GPU sizes (parity-checked against the oracle on sampled
# streams by bench.py itself).  Not bench lines: results go to DESIGN.md section 5.
set -x
timeout 900 python bench.py --streams 4096 --sample-rate 2400000 --nperseg 1024 --window hann --steps 5 --warmup 2 --cpu-streams 16 2>&1 | tail -1 | cut -c1-1500
timeout 900 python bench.py --streams 32768 --sample-rate 2048000 --seconds 0.256 --steps 5 --warmup 2 --cpu-streams 64 2>&1 | tail -1 | cut -c1-1500
timeout 900 python bench.py --streams 1024 --sample-rate 3200000 --nperseg 4096 --trains --steps 5 --warmup 2 --cpu-streams 16 2>&1 | tail -1 | cut -c1-1500
