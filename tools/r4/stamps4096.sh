#!/bin/bash
# per-stage cycle sums of the nperseg-4096 scan step on the config-5 share (tools/variant.sh stamps -DRT_STAMPS): tools/r4/stamps4096.sh <tag>
out=gpurun_out/${1:-r4st}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_var_stamps.so timeout -k 10 300 python3 bench.py --workload config5 --total-streams 1024 --lanes 1 --no-cpu-baseline --steps 3 --warmup 1 --settle 1 --isolated-steps 0 --parity-streams 0 > $out/bench.json 2> $out/run.err || exit 1
grep RT_STAMPS $out/run.err | tail -2 > $out/stamps.txt
cat $out/stamps.txt
