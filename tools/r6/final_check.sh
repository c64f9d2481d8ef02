#!/bin/bash
# round 6, the final tree: the whole -m gpu suite, the bare bench command (the driver's line), the plain parity soak (seed 63) alone
# usage (through gpurun): tools/r6/final_check.sh <tag> [soak seconds]
out=gpurun_out/$1; mkdir -p $out; secs=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $out/tests_all.txt 2>&1; rc=$?; tail -3 $out/tests_all.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $out/bench_bare.json 2> $out/bench_bare.err; rc=$?; python3 tools/show_bench.py $out/bench_bare.json | cut -c1-250; [ $rc -eq 0 ] || exit $rc
timeout -k 10 $((secs + 500)) python3 tests/perf/soak_parity.py $secs 63 > $out/soak_seed63.txt 2>&1; echo "plain soak rc=$?"; tail -3 $out/soak_seed63.txt
