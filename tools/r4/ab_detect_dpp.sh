#!/bin/bash
# detection kernels: DPP / permlane butterflies and scans (default) against ds_bpermute chains (variant prev), one box: kernel stats (Min = a launch with nothing beside it)
# at config 4's eighth and the default geometry, one lane, then the whole path interleaved.   tools/r4/ab_detect_dpp.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in analyze var_prev; do
  for w in "config4 --total-streams 4096" "config5 --total-streams 1024"; do
    n=$(echo $w | cut -d' ' -f1); d=$out/s_${v}_$n
    RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload $w --lanes 1 --steps 8 --warmup 2 --settle 3 --isolated-steps 0 --no-cpu-baseline --parity-streams 0 > $out/bench_${v}_$n.json 2> $out/bench_${v}_$n.err
    echo "== $v $n (kernel, calls, total ns, avg ns, %, min ns, max ns)"; grep "rt::detect\|rt::final" $(ls $d/*/*kernel_stats.csv | head -1) | cut -c1-110
    rm -rf $d
  done
done > $out/kernels.txt
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'detect_ms', r.get('detect_kernel_ms'), 'records', d['config']['records_per_step'])"; }
run() { name=$1; shift; for rep in 1 2; do for v in analyze var_prev; do
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --parity-streams 0 --isolated-steps 0 "$@" 2>>$out/err.txt | line "$name $v rep $rep" >> $out/ab.txt; done; done; }
run config4_eighth --workload config4 --total-streams 4096 --steps 16 --warmup 4 --settle 3
run default_geometry_clean --sample-rate 300000 --streams 4096 --steps 30 --warmup 5 --settle 20
run config5_share --workload config5 --total-streams 1024 --steps 8 --warmup 2 --settle 3
run config3 --workload config3 --steps 8 --warmup 2 --settle 3
cat $out/kernels.txt $out/ab.txt
