#!/bin/bash
# round 6: finalize_records after its counters are read by sixteen lanes at once -- parity tests that exercise it, then kernel durations
# (rocprofv3, one lane) at the reference's defaults and config 4's eighth, then the three-lane whole path
# usage (through gpurun): tools/r6/finalize_check.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_runner.py -m gpu -q -x -k "not fullsize" ) > $out/tests.txt 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 $out/tests.txt; [ $rc -eq 0 ] || exit $rc
common="--steps 12 --warmup 3 --settle 5 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1"
for wl in "defaults --sample-rate 300000 --streams 4096 --nperseg 256" "config4_eighth --workload config4 --total-streams 4096"; do
  name=${wl%% *}; flags=${wl#* }
  d=$out/prof_${name}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common $flags > $out/bench_${name}.json 2> $out/bench_${name}.err || { echo "failed $name"; tail -5 $out/bench_${name}.err; exit 1; }
  echo "== $name"; grep "rt::" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-150
  cp $(ls $d/*/*kernel_stats.csv | head -1) $out/kernel_stats_${name}.csv; rm -rf $d
done
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'], 'parity_bad', (d.get('parity') or {}).get('streams_mismatched'))"; }
common="--steps 40 --warmup 5 --settle 10 --isolated-steps 10 --cpu-streams 4 --parity-streams 8 --other-configs off"
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py $common --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256 2>>$out/err.txt | line "defaults clean" | tee -a $out/bench.txt
  timeout -k 10 300 python3 bench.py $common --lanes 3 --sample-rate 300000 --streams 4096 --nperseg 256 --noise-dbw -88 2>>$out/err.txt | line "defaults floor -88" | tee -a $out/bench.txt
done
timeout -k 10 300 python3 bench.py $common --steps 10 --workload config4 --lanes 1 2>>$out/err.txt | line "config4 all streams" | tee -a $out/bench.txt
