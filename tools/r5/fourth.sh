#!/bin/bash
out=gpurun_out/r5f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
grep -q " passed" $out/tests.txt || exit 1
grep -q "failed" $out/tests.txt && exit 1
export RT_EXP_STREAMS=1
for v in r5p analyze_diag; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so; [ $v = analyze_diag ] && export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_analyze_diag.so
  bash tools/r5/trace_one.sh r5f $v --lanes 1 --mode runfilter --noise-dbw -88 > /dev/null
  echo "$v: $(grep 'stft_scan<1, 6' gpurun_out/r5f/steps_$v.txt | head -1)"
done
unset RT_EXP_STREAMS RT_ANALYZE_LIB
FLOORS="-88 -92" LANES="1 2" bash tools/r5/ab_dg.sh r5f_ab r04 default
