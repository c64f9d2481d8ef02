#!/bin/bash
# MODE 6 ablations: kernel durations with everything in order on one stream (RT_EXP_STREAMS=1), one lane, default geometry floor -88
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RT_EXP_STREAMS=1
for v in r5p e6_1 e6_2 e6_4 e6_8 e6_15 r5p; do
  export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_$v.so
  bash tools/r5/trace_one.sh r5e $v --lanes 1 --mode runfilter --noise-dbw -88 > /dev/null
  echo "$v: $(grep 'stft_scan<1, 6' gpurun_out/r5e/steps_$v.txt | head -1)"
done
bash tools/r5/trace_one.sh r5e clean_sparse --lanes 1 --mode sparse > /dev/null
echo "sparse clean (r5p): $(grep 'stft_scan<1, 0' gpurun_out/r5e/steps_clean_sparse.txt | head -1)"
