#!/bin/bash
# round 6, the last binary: once more, other seeds (75 .. 78): plain, general sizes, NaN-poisoned inputs, BIG
# usage (through gpurun): tools/r6/soak4.sh <tag> <seconds each>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 75 > $out/soak_seed75.txt 2>&1; echo "plain 75 rc=$?"; tail -1 $out/soak_seed75.txt | cut -c1-250; grep -c "error -5" $out/soak_seed75.txt
SOAK_GENERAL=1 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 76 > $out/soak_general_seed76.txt 2>&1; echo "general 76 rc=$?"; tail -1 $out/soak_general_seed76.txt | cut -c1-250; grep -c "error -5" $out/soak_general_seed76.txt
SOAK_POISON=1 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 77 > $out/soak_poison_seed77.txt 2>&1; echo "poison 77 rc=$?"; tail -1 $out/soak_poison_seed77.txt | cut -c1-250
SOAK_BIG=1 timeout -k 10 $((secs + 500)) python3 tests/perf/soak_parity.py $secs 78 > $out/soak_big_seed78.txt 2>&1; echo "big 78 rc=$?"; tail -1 $out/soak_big_seed78.txt | cut -c1-250
