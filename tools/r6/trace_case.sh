#!/bin/bash
# one case of the parity soak with fetch_one's decisions traced (diagnostic library, RT_TRACE_FETCH=1): tools/r6/trace_case.sh <tag> <seed> <case> [env...]
tag=$1; seed=$2; case=$3; shift 3; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
env "$@" RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_analyze_diag.so RT_TRACE_FETCH=1 SOAK_FIRST_CASE=$case timeout -k 10 600 python3 tests/perf/soak_parity.py 0.01 $seed > $out/trace_seed${seed}_case${case}.txt 2>&1
tail -40 $out/trace_seed${seed}_case${case}.txt | cut -c1-250
