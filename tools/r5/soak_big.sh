#!/bin/bash
# big-size randomised parity soak (long buffers, 16 .. 128 streams, many pulses) on the round's final binary: tools/r5/soak_big.sh <tag> <seconds>
out=gpurun_out/$1; mkdir -p $out; secs=${2:-400}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOAK_BIG=1 timeout -k 10 $((secs + 400)) python3 tests/perf/soak_parity.py $secs 54 > $out/soak_big_seed54.txt 2>&1; echo "big rc=$?"; tail -3 $out/soak_big_seed54.txt
