#!/bin/bash
# round 6: sparse detection by groups of candidate lists (detect_group) -- first the parity tests that touch it, then A/B against the
# per-list waves on one box: the reference's defaults clean / under a noise floor / at nperseg 128 (three lanes), config 4 with all streams
# usage (through gpurun): tools/r6/groups_first.sh <tag> [tests|bench|all]
tag=$1; what=${2:-all}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $what != bench ]; then
sel="test_detection_by_groups or test_whole_stream_detection or test_golden_iq_case or test_record_capacity_grows or test_other_baseline_configs_full_geometry or test_thousands_of_plateaus or test_exact_run_length_prefilter_equals_dense or test_look_back_over_several_chunks"
( time timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$sel" ) > $out/tests.txt 2>&1; rc=$?
echo "tests rc=$rc"; tail -15 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
fi
[ $what = tests ] && exit 0
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 40 --warmup 5 --settle 10 --isolated-steps 10 --cpu-streams 4 --parity-streams 8 --other-configs off"
for rep in 1 2; do
for g in off auto; do
  timeout -k 10 300 python3 bench.py $common --group-detect $g --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256 2>>$out/err.txt | line "defaults clean groups $g" | tee -a $out/bench.txt || exit 1
  timeout -k 10 300 python3 bench.py $common --group-detect $g --lanes 1 --sample-rate 300000 --streams 4096 --nperseg 256 2>>$out/err.txt | line "defaults clean one lane groups $g" | tee -a $out/bench.txt || exit 1
  timeout -k 10 300 python3 bench.py $common --group-detect $g --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256 --noise-dbw -88 2>>$out/err.txt | line "defaults floor -88 groups $g" | tee -a $out/bench.txt || exit 1
  timeout -k 10 300 python3 bench.py $common --group-detect $g --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 128 2>>$out/err.txt | line "defaults nperseg 128 groups $g" | tee -a $out/bench.txt || exit 1
  [ $rep != 1 ] || { timeout -k 10 300 python3 bench.py $common --steps 10 --group-detect $g --workload config4 --lanes 1 2>>$out/err.txt | line "config4 all streams groups $g" | tee -a $out/bench.txt || exit 1; }
done
done
