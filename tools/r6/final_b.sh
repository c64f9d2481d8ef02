#!/bin/bash
# round 6, the final tree, part B: the bare command twice more (timed), smoke(), the BIG parity soak and the extract / runner soaks
# usage (through gpurun): tools/r6/final_b.sh <tag> [soak seconds]
tag=$1; out=$PWD/gpurun_out/$tag; mkdir -p $out; secs=${2:-150}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for k in 2 3; do ( time timeout -k 10 400 python3 bench.py > $out/bench_n1_run$k.json 2> $out/bench_n1_run$k.err ) 2>&1 | grep real; python3 tools/show_bench.py $out/bench_n1_run$k.json | cut -c1-330; done
SOAK_BIG=1 timeout -k 10 $((secs + 500)) python3 tests/perf/soak_parity.py $secs 68 > $out/soak_big_seed68.txt 2>&1; echo "big rc=$?"; tail -1 $out/soak_big_seed68.txt | cut -c1-250
timeout -k 10 $((secs + 300)) python3 tests/perf/soak_extract.py 60 69 > $out/soak_extract_seed69.txt 2>&1; echo "extract rc=$?"; tail -1 $out/soak_extract_seed69.txt | cut -c1-250
timeout -k 10 $((secs + 300)) python3 tests/perf/soak_runner.py 60 70 > $out/soak_runner_seed70.txt 2>&1; echo "runner rc=$?"; tail -1 $out/soak_runner_seed70.txt | cut -c1-250
