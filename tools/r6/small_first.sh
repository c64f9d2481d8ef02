#!/bin/bash
# round 6, first GPU run of the fused scans at nperseg 32 / 64 / 128: the parity tests that touch them, then throughput next to nperseg 256
# usage (through gpurun): tools/r6/small_first.sh <tag>
tag=${1:-r6a}; what=${2:-all}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $what != bench ]; then
sel="test_spectrogram_matches_oracle or test_spectrogram_into_a_map or test_golden_iq_case or test_batch_of_streams_matches_oracle or test_look_back_over_several_chunks or test_uint8_wire_format_ingestion or test_detrend_by_linearity or test_exact_run_length_prefilter_equals_dense or test_other_powers_of_two or test_small_sizes_take_any or test_unsupported_nperseg or test_run_length_prefilter_equals_dense or test_lanes_give_the_same_records"
( time timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$sel" ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -25 $out/tests.txt
fi
[ $what = tests ] && exit 0
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'L', d['config']['segments_per_chunk'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 20 --warmup 5 --settle 10 --isolated-steps 10 --cpu-streams 4 --parity-streams 8 --other-configs off"
for n in 128 64 32 256; do
  for lanes in 1 3; do
    timeout -k 10 300 python3 bench.py $common --lanes $lanes --sample-rate 300000 --streams 4096 --nperseg $n 2>>$out/err.txt | line "defaults nperseg $n lanes $lanes" | tee -a $out/bench.txt
  done
done
timeout -k 10 300 python3 bench.py $common --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 128 --mode dense 2>>$out/err.txt | line "defaults nperseg 128 dense" | tee -a $out/bench.txt
timeout -k 10 300 python3 bench.py $common --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 128 --input u8 2>>$out/err.txt | line "defaults nperseg 128 uint8 lanes 3" | tee -a $out/bench.txt
timeout -k 10 300 python3 bench.py $common --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 128 --noise-dbw -88 2>>$out/err.txt | line "defaults nperseg 128 floor -88 lanes 3" | tee -a $out/bench.txt
