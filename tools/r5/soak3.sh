#!/bin/bash
# randomised parity soak on the small general sizes only (stft_small): tools/r5/soak3.sh <tag> <seconds> <seed>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}; seed=${3:-71}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_GENERAL=1 SOAK_GENERAL_SIZES=32,64,128 timeout -k 10 $((secs + 200)) python3 tests/perf/soak_parity.py $secs $seed > $out/soak_small_seed$seed.txt 2>&1; echo "small rc=$?"; tail -4 $out/soak_small_seed$seed.txt
