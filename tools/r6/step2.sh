#!/bin/bash
# round 6: (1) fences between the pairs of stft_wg's passes, A/B on one box (8192 and 16384); (2) the sharded_configs test; (3) N = 1 bare bench line
tag=${1:-r6e}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'mode', d['config']['mode'], 'L', d['config']['segments_per_chunk'], 'records', d['config']['records_per_step'])"; }
common="--steps 10 --warmup 3 --settle 4 --isolated-steps 5 --no-cpu-baseline --parity-streams 0 --other-configs off --lanes 1 --workload config5 --total-streams 512"
for rep in 1 2; do
  for v in product wg_nofence; do
    if [ $v = product ]; then unset RT_ANALYZE_LIB; else export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so; fi
    for n in 8192 16384; do
      timeout -k 10 300 python3 bench.py $common --nperseg $n 2>>$out/err.txt | line "$n $v" | tee -a $out/ab.txt
    done
  done
done
unset RT_ANALYZE_LIB
( time timeout -k 10 900 python -m pytest tests/test_multigpu.py -m gpu -q -x -k "sharded_configs or other_configs_block" ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -5 $out/tests.txt
( time timeout -k 10 600 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err ); echo "bench rc=$?"
python3 tools/show_bench.py $out/bench_n1.json 2>/dev/null | head -60 || head -c 3000 $out/bench_n1.json
