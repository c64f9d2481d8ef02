#!/bin/bash
# stft_small (nperseg 32 / 64 / 128): parity tests of the general sizes, then throughput next to the fused dense scan at 256
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_abi.py -x -q -m gpu -k "spectrogram_matches or other_powers or golden_iq_case or unsupported or abi" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 4 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1"
for n in 128 64 32; do
python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg $n 2>/dev/null | line "defaults nperseg $n" | tee -a $out/bench.txt
done
python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg 256 --mode dense 2>/dev/null | line "defaults nperseg 256 dense (fused)" | tee -a $out/bench.txt
