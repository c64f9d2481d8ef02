#!/bin/bash
# round 6, the final binary: (1) tools/profile_round.sh (kernel stats of the default command, three lanes and one; PMC traffic json);
# (2) kernel stats of the sizes that became fused scans (nperseg 128 at the reference's defaults, 8192 at 3.2 MS/s), one lane;
# (3) the bare N = 1 line; (4) the sharded block at world 1 with the full populations
tag=${1:-r6k}; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/profile_round.sh $tag > $out/profile_round.txt 2>&1; echo "profile_round rc=$?"; tail -3 $out/profile_round.txt | cut -c1-300
common="--lanes 1 --no-cpu-baseline --steps 10 --warmup 2 --settle 4 --isolated-steps 0 --other-configs off --parity-streams 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s128 -- python3 bench.py $common --sample-rate 300000 --streams 4096 --nperseg 128 > $out/n128_one_lane.json 2> $out/n128.err
cp $(ls $out/s128/*/*kernel_stats.csv | head -1) $out/n128_defaults_one_lane_kernel_stats.csv; rm -rf $out/s128
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s8192 -- python3 bench.py $common --workload config5 --total-streams 512 --nperseg 8192 > $out/n8192_one_lane.json 2> $out/n8192.err
cp $(ls $out/s8192/*/*kernel_stats.csv | head -1) $out/n8192_one_lane_kernel_stats.csv; rm -rf $out/s8192
grep -h "stft_scan\|stft_wg" $out/n128_defaults_one_lane_kernel_stats.csv $out/n8192_one_lane_kernel_stats.csv | cut -c1-150
( time timeout -k 10 600 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err ); echo "bench rc=$?"
python3 tools/show_bench.py $out/bench_n1.json | cut -c1-400
( time timeout -k 10 600 python3 bench.py --sharded-configs on --other-configs off --no-cpu-baseline --parity-streams 4 > $out/bench_n1_sharded_block.json 2> $out/bench_n1_sharded.err ); echo "sharded rc=$?"
python3 tools/show_bench.py $out/bench_n1_sharded_block.json | cut -c1-400
