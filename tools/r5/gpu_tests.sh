#!/bin/bash
# the whole -m gpu suite on the box, output kept under gpurun_out/r5m (tools/r5/gpu_tests.sh)
out=gpurun_out/r5m; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -30 $out/tests.txt
