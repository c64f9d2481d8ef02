#!/bin/bash
# round 6, the last tree: more randomised parity soaks (fresh seeds): plain, general sizes, NaN-poisoned inputs, BIG
# usage (through gpurun): tools/r6/soak4.sh <tag> <seconds each>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 71 > $out/soak_seed71.txt 2>&1; echo "plain 71 rc=$?"; tail -1 $out/soak_seed71.txt | cut -c1-250; grep -c "error -5" $out/soak_seed71.txt
SOAK_GENERAL=1 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 72 > $out/soak_general_seed72.txt 2>&1; echo "general 72 rc=$?"; tail -1 $out/soak_general_seed72.txt | cut -c1-250; grep -c "error -5" $out/soak_general_seed72.txt
SOAK_POISON=1 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 73 > $out/soak_poison_seed73.txt 2>&1; echo "poison 73 rc=$?"; tail -1 $out/soak_poison_seed73.txt | cut -c1-250
SOAK_BIG=1 timeout -k 10 $((secs + 500)) python3 tests/perf/soak_parity.py $secs 74 > $out/soak_big_seed74.txt 2>&1; echo "big 74 rc=$?"; tail -1 $out/soak_big_seed74.txt | cut -c1-250
