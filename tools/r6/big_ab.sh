#!/bin/bash
# round 6: stft_wg at nperseg 8192 -- the product against timing-only ablations (tools/variant.sh wg_a<mask> -DRT_WG_ABL=<mask>), same box
# usage (through gpurun): tools/r6/big_ab.sh <tag> [variants...]
tag=${1:-r6c}; shift; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(test_spectrogram_matches_oracle and (8192 or 16384)) or (test_golden_iq_case and n8192) or (test_batch_of_streams_matches_oracle and (8192 or 16384)) or (test_look_back_over_several_chunks and 8192) or (test_uint8_wire_format_ingestion and 8192)" ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'L', d['config']['segments_per_chunk'], 'records', d['config']['records_per_step'])"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 5 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1 --workload config5 --total-streams 512 --nperseg 8192"
for rep in 1 2; do
  for v in product "$@"; do
    if [ $v = product ]; then unset RT_ANALYZE_LIB; else export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so; fi
    timeout -k 10 300 python3 bench.py $common 2>>$out/err.txt | line "8192 $v" | tee -a $out/ab.txt
  done
done
unset RT_ANALYZE_LIB
timeout -k 10 300 python3 bench.py $common --mode dense 2>>$out/err.txt | line "8192 product dense" | tee -a $out/ab.txt
timeout -k 10 300 python3 bench.py $common --nperseg 16384 2>>$out/err.txt | line "16384 product" | tee -a $out/ab.txt
timeout -k 10 300 python3 bench.py $common --window blackmanharris 2>>$out/err.txt | line "8192 product blackmanharris (window table)" | tee -a $out/ab.txt
