#!/bin/bash
# nperseg 4096: the one-wave-per-segment scan (default build) against the four-waves-per-segment one (variant old4096,
# tools/variant.sh old4096 -DRT_WAVE64_4096=0), same box: transform round-off, the config-5 share (one lane), chunk lengths.
#   tools/r4/ab4096.sh <tag> [segs-per-chunk values...]
tag=${1:-r4ab}; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
old=$PWD/pyradiotracking_amd/librt_var_old4096.so
python3 tests/perf/fft_round_off.py 4096 > $out/accuracy.txt 2>&1 || exit 1
[ -f $old ] && { RT_ANALYZE_LIB=$old python3 tests/perf/fft_round_off.py 4096 >> $out/accuracy.txt 2>&1 || exit 1; }
common="--workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 10 --warmup 3 --settle 4 --isolated-steps 10"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', r['kernel_ms'], 'frac', r['frac'], 'detect_ms', r['detect_kernel_ms'], 'records', d['config']['records_per_step'], 'parity', d.get('parity',{}).get('streams_mismatched'), '/', d.get('parity',{}).get('streams_checked'))"; }
timeout -k 10 300 python3 bench.py $common 2>>$out/err.txt | line "wave64 default-L" >> $out/ab.txt || exit 1
[ -f $old ] && { RT_ANALYZE_LIB=$old timeout -k 10 300 python3 bench.py $common 2>>$out/err.txt | line "old4096 default-L" >> $out/ab.txt || exit 1; }
for L in "$@"; do
  timeout -k 10 300 python3 bench.py $common --segs-per-chunk $L --parity-streams 4 2>>$out/err.txt | line "wave64 L=$L" >> $out/ab.txt || exit 1
done
cat $out/accuracy.txt $out/ab.txt
