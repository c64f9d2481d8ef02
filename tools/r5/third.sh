#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_r5p.so
for p in high normal low; do
  RT_EXP_TAIL_PRIO=$p bash tools/r5/trace_one.sh r5d noisy_l1_$p --lanes 1 --mode runfilter --noise-dbw -88
done
unset RT_ANALYZE_LIB
FLOORS="-88" LANES=1 bash tools/r5/ab_dg.sh r5d_ab r04 r5p:RT_EXP_TAIL_PRIO=high r5p:RT_EXP_TAIL_PRIO=normal r5p:RT_EXP_TAIL_PRIO=low r5p:RT_EXP_STREAMS=1
