#!/bin/bash
# round 6, the very last tree: the whole -m gpu suite, smoke(), the bare command twice (timed)
# usage (through gpurun): tools/r6/final_c.sh <tag>
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python3 -m pytest tests -m gpu -q -x ) > $out/tests_all.txt 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 $out/tests_all.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for k in 1 2; do ( time timeout -k 10 400 python3 bench.py > $out/bench_n1_run$k.json 2> $out/bench_n1_run$k.err ) 2>&1 | grep real; python3 tools/show_bench.py $out/bench_n1_run$k.json | cut -c1-330; done
