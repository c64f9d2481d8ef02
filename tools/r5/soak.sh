#!/bin/bash
# randomised parity soaks on the round's final binary: tools/r5/soak.sh <tag> <seconds each>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-240}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_GENERAL=1 timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs ${3:-51} > $out/soak_general_seed${3:-51}.txt 2>&1; echo "general rc=$?"; tail -4 $out/soak_general_seed${3:-51}.txt
timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs ${4:-52} > $out/soak_seed${4:-52}.txt 2>&1; echo "plain rc=$?"; tail -4 $out/soak_seed${4:-52}.txt
