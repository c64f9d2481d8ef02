for v in default dab1 dab2 dab3; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  RT_ANALYZE_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --isolated-steps 0 --steps 10 --warmup 3 --settle 3 --lanes 1 --workload config4 --total-streams 8192 --parity-streams 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['kernel_ms'], 'detect_ms', d['roofline']['detect_kernel_ms'], 'records', d['config']['records_per_step'])"
done
