#!/bin/bash
# Round 3, exploration 2: timing-only diagnostic builds of the nperseg 4096 / 1024 scan kernels (window loads, barriers)
# + the new record-pool tests on the product build.
out=gpurun_out/r3b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "4096 3200000 512" "1024 2400000 1024"; do
  for v in analyze var_nowin var_nobar0 var_nobar1 var_nobar01 var_nowinbar ablate_1 var_nowin_a1 var_nobar0_a1; do
    RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 200 python tools/ablate_large.py $g 2>>$out/ablate.err | tail -1 >> $out/ablate.txt
  done
  echo "ablation $g done"
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or truncated or many_records" > $out/pytest_pool.txt 2>&1; echo "pytest rc $?"
tail -5 $out/pytest_pool.txt
