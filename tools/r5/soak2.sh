#!/bin/bash
# randomised parity soak with the general sizes (powers of two and not) on the round's final binary: tools/r5/soak2.sh <tag> <seconds>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-300}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_GENERAL=1 timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs 53 > $out/soak_general_seed53.txt 2>&1; echo "general rc=$?"; tail -3 $out/soak_general_seed53.txt
python3 bench.py --steps 10 --warmup 3 --settle 4 --isolated-steps 4 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 300 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defaults nperseg 300 (Bluestein): value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'parity_bad', d['parity']['streams_mismatched'])"
python3 bench.py --steps 10 --warmup 3 --settle 4 --isolated-steps 4 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1 --sample-rate 2400000 --streams 512 --nperseg 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2.4 MS/s nperseg 1000 (Bluestein): value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'parity_bad', d['parity']['streams_mismatched'])"
