#!/bin/bash
# the -m gpu suite in one process + the default bench line (driver flags) -> gpurun_out/<tag>/
tag=${1:-suite}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; rc=$?
tail -15 $out/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || exit 1
python tools/show_bench.py $out/bench_default.json
