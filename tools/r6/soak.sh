#!/bin/bash
# randomised parity soaks (tests/perf/soak_parity.py): the sizes that became fused scans this round, the general mix, the plain mix
# usage (through gpurun): tools/r6/soak.sh <tag> <seconds each>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_GENERAL=1 SOAK_GENERAL_SIZES=32,64,128,128,8192,16384 timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs 61 > $out/soak_new_fused_sizes_seed61.txt 2>&1; echo "new sizes rc=$?"; tail -3 $out/soak_new_fused_sizes_seed61.txt
SOAK_GENERAL=1 timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs 62 > $out/soak_general_seed62.txt 2>&1; echo "general rc=$?"; tail -3 $out/soak_general_seed62.txt
timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs 63 > $out/soak_seed63.txt 2>&1; echo "plain rc=$?"; tail -3 $out/soak_seed63.txt
