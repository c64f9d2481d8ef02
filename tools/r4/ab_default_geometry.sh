#!/bin/bash
# the reference's default geometry (4 096 streams x 300 kS/s, nperseg 256), same box: chunk length 32 against the handle's own
# choice, clean input (sparse level) and a floor 2 dB over the threshold (exact pre-filter), one and two lanes; optional library variants
#   tools/r4/ab_default_geometry.sh <tag> [variant...]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--sample-rate 300000 --streams 4096 --steps 30 --warmup 5 --settle 20 --isolated-steps 0 --no-cpu-baseline --parity-streams 0"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: value', d['value'], 'ms/step', d['ms_per_step'], 'mode', d['config']['mode'], 'records', d['config']['records_per_step'])"; }
for v in default "$@"; do
  lib=$PWD/pyradiotracking_amd/librt_var_$v.so; [ "$v" = default ] && lib=$PWD/pyradiotracking_amd/librt_analyze.so
  for lanes in 1 2; do
    for extra in "" "--segs-per-chunk 32"; do
      RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py $common --lanes $lanes $extra 2>>$out/err.txt | line "$v clean lanes $lanes $extra" >> $out/ab.txt
      RT_ANALYZE_LIB=$lib timeout -k 10 300 python3 bench.py $common --lanes $lanes $extra --noise-dbw -88 --mode runfilter 2>>$out/err.txt | line "$v floor-88 runfilter lanes $lanes $extra" >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
