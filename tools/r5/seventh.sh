#!/bin/bash
out=gpurun_out/r5i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
grep -q " passed" $out/tests.txt || exit 1
grep -q "failed" $out/tests.txt && exit 1
FLOORS="-92 -90 -88 -86" LANES="1 2" bash tools/r5/ab_dg.sh r5i_ab r04 default
