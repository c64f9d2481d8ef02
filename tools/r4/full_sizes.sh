#!/bin/bash
# every BASELINE config at its one-GPU size on the round's sources (bench.py checks 16 sampled streams against the oracle):
#   default line with the driver's flags, config 3, config 4 (all 32 768 streams at B = 524 288 if they fit, else the 8 192-stream share),
#   config 5 (the 1 024-stream share and all 8 192 streams on one GPU), uint8 at config 2.   tools/r4/full_sizes.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 900 python3 bench.py "$@" 2>>$out/err.txt | tail -1 > $out/$name.json || echo "{\"failed\": \"$name\"}" > $out/$name.json; python3 tools/show_bench.py $out/$name.json 2>/dev/null | cut -c1-330; }
run default --steps 20 --warmup 5
run config3 --workload config3 --steps 8 --warmup 2 --settle 3 --isolated-steps 5 --cpu-streams 16
run config4_share --workload config4 --total-streams 8192 --steps 8 --warmup 2 --settle 3 --isolated-steps 5 --cpu-streams 64
run config5_share --workload config5 --total-streams 1024 --steps 8 --warmup 2 --settle 3 --isolated-steps 5 --cpu-streams 16
run config2_u8 --input u8 --steps 20 --warmup 5 --no-cpu-baseline
run config5_all --workload config5 --steps 4 --warmup 1 --settle 2 --isolated-steps 3 --no-cpu-baseline
run config4_all --workload config4 --steps 4 --warmup 1 --settle 2 --isolated-steps 3 --no-cpu-baseline
