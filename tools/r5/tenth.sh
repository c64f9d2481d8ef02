#!/bin/bash
out=gpurun_out/r5n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $out/tests.txt
grep -q " passed" $out/tests.txt || exit 1
grep -q "failed" $out/tests.txt && exit 1
bash tools/profile_round.sh r5a > $out/profile_round.log 2>&1
tail -5 $out/profile_round.log | cut -c1-300
cat gpurun_out/prof_r5a/pmc_traffic.json
