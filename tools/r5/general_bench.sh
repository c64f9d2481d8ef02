#!/bin/bash
# throughput of the general transform's path (dense) at nperseg 128 / 64 / 8192 / 16384 next to the fused kernels' dense path at 256 / 4096
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 4 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1"
python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg 128 2>/dev/null | line "defaults nperseg 128 (general)"
python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg 64 2>/dev/null | line "defaults nperseg 64 (general)"
python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg 256 --mode dense 2>/dev/null | line "defaults nperseg 256 dense (fused)"
python3 bench.py $common --workload config5 --total-streams 512 --nperseg 8192 2>/dev/null | line "3.2 MS/s nperseg 8192 (general)"
python3 bench.py $common --workload config5 --total-streams 512 --nperseg 16384 2>/dev/null | line "3.2 MS/s nperseg 16384 (general)"
python3 bench.py $common --workload config5 --total-streams 512 --mode dense 2>/dev/null | line "3.2 MS/s nperseg 4096 dense (fused)"
