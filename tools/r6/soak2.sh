#!/bin/bash
# round 6, after the whole-stream detection: randomised parity soaks with that form forced in half of the cases at nperseg <= 256
# (tests/perf/soak_parity.py): the plain mix from case 28 of seed 63 (case 27 is 28 noisy nperseg-2048 streams with snr 0 -- hours of oracle),
# a fresh plain seed, the small fused sizes, the BIG mix
# usage (through gpurun): tools/r6/soak2.sh <tag> <seconds each>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_FIRST_CASE=28 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 63 > $out/soak_seed63_from_case28.txt 2>&1; echo "plain 63 rc=$?"; tail -2 $out/soak_seed63_from_case28.txt | cut -c1-250
timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 64 > $out/soak_seed64.txt 2>&1; echo "plain 64 rc=$?"; tail -2 $out/soak_seed64.txt | cut -c1-250
SOAK_GENERAL=1 SOAK_GENERAL_SIZES=32,64,128,128,256 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 65 > $out/soak_small_sizes_seed65.txt 2>&1; echo "small rc=$?"; tail -2 $out/soak_small_sizes_seed65.txt | cut -c1-250
