#!/bin/bash
out=gpurun_out/r5b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
[ "$(grep -c passed $out/tests.txt)" -ge 1 ] || exit 1
grep -q failed $out/tests.txt && exit 1
bash tools/r5/ab_dg.sh r5b_ab r04 default
