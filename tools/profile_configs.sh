#!/bin/bash
# rocprofv3 --kernel-trace --stats of BASELINE configs 3 and 5 (one GPU's population, ONE lane: every stft_scan row of the
# csv is one launch over all streams) and SQ counter passes (LDS bank conflicts) of the nperseg 1024 / 4096 kernels.
# usage (on the GPU box through gpurun): tools/profile_configs.sh <tag>   -> gpurun_out/prof_<tag>/
tag=${1:-cfg}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
common="--lanes 1 --no-cpu-baseline --steps 6 --warmup 2 --settle 4 --isolated-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s3 -- python3 bench.py --workload config3 $common > $out/config3_one_lane.json 2> $out/config3.err
cp $(ls $out/s3/*/*kernel_stats.csv | head -1) $out/config3_one_lane_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s5 -- python3 bench.py --workload config5 --total-streams 1024 $common > $out/config5_one_lane.json 2> $out/config5.err
cp $(ls $out/s5/*/*kernel_stats.csv | head -1) $out/config5_one_lane_kernel_stats.csv
# (round 6: at nperseg >= 1024 the detection runs in order on the scan's stream in the product -- rt_create: `second = R3 <= 2` -- so every stft_scan64
# row above IS a launch with nothing beside it; the former second pass under RT_EXP_ONE_STREAM profiled the same configuration twice)
rm -rf $out/s3 $out/s5
tools/sq_counters.sh ${tag}_n1024 1024 2400000 128 > /dev/null
tools/sq_counters.sh ${tag}_n4096 4096 3200000 128 > /dev/null
grep -h "stft_scan" $out/config3_one_lane_kernel_stats.csv $out/config5_one_lane_kernel_stats.csv | cut -c1-140
grep -h "LDS_BANK_CONFLICT\|LDS_IDX_ACTIVE" gpurun_out/sq_${tag}_n1024.txt gpurun_out/sq_${tag}_n4096.txt | grep "0, false"
