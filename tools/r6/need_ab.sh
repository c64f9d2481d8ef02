#!/bin/bash
# round 6: the exact pre-filter's planner writing only the need words that keep anything (the listed scan zeroes what it reads) against the
# library before (pyradiotracking_amd/librt_var_before.so), one box: parity tests of the pre-filter, kernel durations (one lane), whole path
# (three lanes) at the reference's defaults with the floor 2 dB over the threshold, complex64 and uint8
# usage (through gpurun): tools/r6/need_ab.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prefilter or runfilter or floor or golden_iq_case or auto or every_mode or noise" ) > $out/tests.txt 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
common="--steps 12 --warmup 3 --settle 8 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 256 --noise-dbw -88"
for v in var_before analyze; do
  d=$out/prof_$v
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common > $out/bench_$v.json 2> $out/bench_$v.err || { echo "failed $v"; tail -5 $out/bench_$v.err; exit 1; }
  echo "== $v"; grep "rt::plan_runs\|stft_scan<1, 7\|stft_scan<1, 6" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-150
  rm -rf $d
done
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 40 --warmup 5 --settle 20 --isolated-steps 0 --cpu-streams 4 --parity-streams 8 --other-configs off --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256"
for rep in 1 2 3; do
for v in var_before analyze; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so
  timeout -k 10 300 python3 bench.py $common --noise-dbw -88 2>>$out/err.txt | line "defaults floor -88 $v" | tee -a $out/bench.txt
  timeout -k 10 300 python3 bench.py $common --input u8 --threshold-dbw -91 2>>$out/err.txt | line "defaults uint8 $v" | tee -a $out/bench.txt
done
done
