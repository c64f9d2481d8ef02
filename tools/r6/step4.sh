#!/bin/bash
# round 6: host sinks by thread count (64 cores), the uint8 path's VALU roofline (PMC), the bench tests that read host_sinks
tag=${1:-r6h}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/r6/sinks.py > $out/sinks.txt 2>&1; echo "sinks rc=$?"; cat $out/sinks.txt
tools/r6/u8_valu.sh $tag > $out/u8_valu.txt 2>&1; echo "u8 valu rc=$?"; tail -12 $out/u8_valu.txt
( time timeout -k 10 600 python -m pytest tests/test_multigpu.py tests/test_consume.py tests/test_match.py tests/test_runner.py -m gpu -q -x ) > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $out/tests.txt
