#!/bin/bash
out=gpurun_out/r5m; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -30 $out/tests.txt
