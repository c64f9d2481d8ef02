#!/bin/bash
out=gpurun_out/r5g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$1.so
( time timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prefilter or runfilter or floor or auto" ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
grep -q " passed" $out/tests.txt || exit 1
grep -q "failed" $out/tests.txt && exit 1
export RT_EXP_STREAMS=1
for v in r5p $1 r5p $1; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so
  bash tools/r5/trace_one.sh r5g $v --lanes 1 --mode runfilter --noise-dbw -88 > /dev/null
  echo "$v: $(grep 'stft_scan<1, 6' gpurun_out/r5g/steps_$v.txt | head -1)"
done
