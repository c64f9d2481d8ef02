#!/bin/bash
# the LDS transforms after batching (nperseg 8192 / 16384, Bluestein sizes): parity tests, then throughput
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_abi.py -x -q -m gpu -k "spectrogram_matches or other_powers or golden_iq_case or unsupported or abi" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 4 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1"
python3 bench.py $common --workload config5 --total-streams 512 --nperseg 8192 2>/dev/null | line "3.2 MS/s nperseg 8192" | tee -a $out/bench.txt
python3 bench.py $common --workload config5 --total-streams 512 --nperseg 16384 2>/dev/null | line "3.2 MS/s nperseg 16384" | tee -a $out/bench.txt
python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg 300 2>/dev/null | line "defaults nperseg 300 (Bluestein, M 1024)" | tee -a $out/bench.txt
python3 bench.py $common --sample-rate 2400000 --streams 512 --nperseg 1000 2>/dev/null | line "2.4 MS/s nperseg 1000 (Bluestein, M 2048)" | tee -a $out/bench.txt
python3 bench.py $common --sample-rate 2400000 --streams 512 --nperseg 1500 2>/dev/null | line "2.4 MS/s nperseg 1500 (Bluestein, M 4096)" | tee -a $out/bench.txt
python3 bench.py $common --workload config5 --total-streams 256 --nperseg 3000 2>/dev/null | line "3.2 MS/s nperseg 3000 (Bluestein, M 8192)" | tee -a $out/bench.txt
python3 bench.py $common --workload config5 --total-streams 256 --nperseg 6000 2>/dev/null | line "3.2 MS/s nperseg 6000 (Bluestein, M 16384)" | tee -a $out/bench.txt
