#!/bin/bash
# round 6: record capacity that grows (tests), then the whole -m gpu suite
tag=${1:-r6f}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "record_capacity_grows or thousands_of_plateaus or truncated_extract or record_pool_grows or pool_that_cannot_grow or many_records_per_stream" ) > $out/tests_cap.txt 2>&1
echo "cap tests rc=$?"; tail -15 $out/tests_cap.txt
( time timeout -k 10 1000 python -m pytest tests -m gpu -q -x ) > $out/tests_all.txt 2>&1
echo "all tests rc=$?"; tail -15 $out/tests_all.txt
