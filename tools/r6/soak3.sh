#!/bin/bash
# round 6: soaks after the capacity fixes (no AUTO case may be delivered truncated): the traced case, the small sizes again (seed 65), a fresh plain seed
# usage (through gpurun): tools/r6/soak3.sh <tag> <seconds each>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_GENERAL=1 SOAK_GENERAL_SIZES=32,64,128,128,256 SOAK_FIRST_CASE=226 timeout -k 10 300 python3 tests/perf/soak_parity.py 0.01 65 2>&1 | grep "^case\|^SOAK" | cut -c1-250
SOAK_GENERAL=1 SOAK_GENERAL_SIZES=32,64,128,128,256 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 65 > $out/soak_small_sizes_seed65.txt 2>&1; echo "small rc=$?"; tail -1 $out/soak_small_sizes_seed65.txt | cut -c1-250; grep -c "error -5" $out/soak_small_sizes_seed65.txt
timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 66 > $out/soak_seed66.txt 2>&1; echo "plain 66 rc=$?"; tail -1 $out/soak_seed66.txt | cut -c1-250; grep "error -5" $out/soak_seed66.txt | cut -c1-200
SOAK_GENERAL=1 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 67 > $out/soak_general_seed67.txt 2>&1; echo "general 67 rc=$?"; tail -1 $out/soak_general_seed67.txt | cut -c1-250; grep "error -5" $out/soak_general_seed67.txt | cut -c1-200
