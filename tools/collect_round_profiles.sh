#!/bin/bash
# copy what tools/profile_round.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/ (tracked) under the round's names
# usage: tools/collect_round_profiles.sh <tag> <prefix, e.g. r02_a>
P=gpurun_out/prof_$1; pre=$2
cp $P/kernel_stats.csv profiles/${pre}_kernel_stats_bench_two_lanes.csv
tail -1 $P/bench.json > profiles/${pre}_bench_output_under_rocprof_two_lanes.json
cp $P/kernel_stats_one_lane.csv profiles/${pre}_kernel_stats_bench_one_lane.csv
tail -1 $P/bench_one_lane.json > profiles/${pre}_bench_output_under_rocprof_one_lane.json
cp $P/pmc_traffic.json profiles/pmc_traffic.json
cp $P/pmc_summary.txt profiles/${pre}_pmc_counters_raw.txt
head -2 profiles/${pre}_kernel_stats_bench_two_lanes.csv | cut -c1-140; head -2 profiles/${pre}_kernel_stats_bench_one_lane.csv | cut -c1-140
python tools/show_bench.py profiles/${pre}_bench_output_under_rocprof_two_lanes.json; python tools/show_bench.py profiles/${pre}_bench_output_under_rocprof_one_lane.json
python -c "
import sys; sys.path.insert(0,'.')
import bench, json; print('pmc hash matches the built scan kernel:', bench.scan_kernel_sha256() == json.load(open('profiles/pmc_traffic.json'))['scan_kernel_sha256'])"
