#!/bin/bash
# Round profile set (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run (three lanes; without the other_configs block and the one-lane pass after the timed
#      region, so that every stft_scan row of the csv is an 85- or 86-stream launch)  -> kernel_stats.csv, bench.json
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/profile_traffic.py -> pmc_summary.txt
# The program itself follows `--` (no shell wrappers under the profiler).
tag=${1:-round}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --isolated-steps 0 --other-configs off > $out/bench.json 2> $out/bench.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 tools/profile_traffic.py > $out/traffic_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 tools/profile_traffic.py > $out/traffic_write.json 2> $out/pmc_write.err
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write > $out/pmc_summary.txt
# corrected bytes per launch + sha256 of the kernel sources -> what bench.py quotes as roofline.traffic
python3 tools/pmc_traffic_json.py $out/pmc_fetch $out/pmc_write $out/traffic_fetch.json $out/pmc_traffic.json "gpurun_out/prof_$tag (tools/profile_round.sh)" > /dev/null
# the same bench in one-lane mode under the profiler: the isolated-launch figure from a kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 bench.py --lanes 1 --no-cpu-baseline --other-configs off > $out/bench_one_lane.json 2> $out/bench_one_lane.err
cp $(ls $out/stats1/*/*kernel_stats.csv | head -1) $out/kernel_stats_one_lane.csv
rm -rf $out/stats1
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
tail -1 $out/bench.json | cut -c1-600; cat $out/kernel_stats.csv | cut -c1-160; cat $out/pmc_summary.txt; tail -1 $out/traffic_fetch.json
