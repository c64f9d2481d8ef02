#!/bin/bash
# round 6, the final tree, part A: the whole -m gpu suite, then tools/r6/final_profiles.sh (kernel stats three lanes / one lane, PMC traffic,
# nperseg 128 / 8192 stats, the bare N = 1 line, the sharded block at world 1)
# usage (through gpurun): tools/r6/final_a.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python3 -m pytest tests -m gpu -q -x ) > $out/tests_all.txt 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 $out/tests_all.txt; [ $rc -eq 0 ] || exit $rc
tools/r6/final_profiles.sh $tag
