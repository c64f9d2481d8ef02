#!/bin/bash
# round 6, first GPU run of stft_wg (nperseg 8192 / 16384, one workgroup per segment): parity tests, then throughput
# usage (through gpurun): tools/r6/big_first.sh <tag> [tests|bench|all]
tag=${1:-r6b}; what=${2:-all}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $what != bench ]; then
sel="(test_spectrogram_matches_oracle and (8192 or 16384)) or (test_golden_iq_case and n8192) or (test_batch_of_streams_matches_oracle and (8192 or 16384)) or (test_look_back_over_several_chunks and 8192) or (test_uint8_wire_format_ingestion and 8192) or test_other_powers_of_two or test_unsupported_nperseg"
( time timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$sel" ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -25 $out/tests.txt
fi
[ $what = tests ] && exit 0
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'L', d['config']['segments_per_chunk'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 5 --cpu-streams 4 --parity-streams 4 --other-configs off --lanes 1"
for n in 8192 16384; do
  for mode in auto dense; do
    timeout -k 10 300 python3 bench.py $common --workload config5 --total-streams 512 --nperseg $n --mode $mode 2>>$out/err.txt | line "3.2 MS/s 512 streams nperseg $n $mode" | tee -a $out/bench.txt
  done
done
timeout -k 10 300 python3 bench.py $common --workload config5 --total-streams 512 --nperseg 8192 --input u8 2>>$out/err.txt | line "3.2 MS/s 512 streams nperseg 8192 uint8" | tee -a $out/bench.txt
