#!/bin/bash
# randomised parity soaks on the final round-3 sources: tools/r3/soak.sh <tag> <seconds> [seed-a seed-big] -> gpurun_out/<tag>/soak_*.log
tag=${1:-r3soak}; secs=${2:-150}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 $((secs + 120)) python tests/perf/soak_parity.py $secs ${3:-31} > $out/soak_a.log 2>&1 || { tail -5 $out/soak_a.log; exit 1; }
tail -3 $out/soak_a.log
SOAK_BIG=1 timeout -k 10 $((secs + 200)) python tests/perf/soak_parity.py $secs ${4:-32} > $out/soak_big.log 2>&1 || { tail -5 $out/soak_big.log; exit 1; }
tail -3 $out/soak_big.log
