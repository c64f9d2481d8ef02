#!/bin/bash
# HBM traffic of the exact run-length pre-filter level (RT_MODE_RUNFILTER: bits scan + plan_runs + listed scan) at the
# reference's default geometry (300 kS/s, nperseg 256, 8 - 40 ms) with the noise floor AT the threshold (sigma 1e-5 at
# 300 kS/s = -151.8 dBW per bin; pulses 20 .. 34 dB over it): two --pmc passes.
# usage (GPU box): tools/profile_runfilter_traffic.sh <tag> [threshold dBW, default -151.8 = the floor; -153.8 puts the floor 2 dB over it]
#   -> gpurun_out/prof_<tag>/runfilter_pmc_summary_thr<threshold>.txt
tag=${1:-rf}; thr=${2:--151.8}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export RT_PROF_MODE=runfilter RT_PROF_FS=300000 RT_PROF_STREAMS=2048 RT_PROF_THRESHOLD_DBW=$thr RT_PROF_PEAK_DBW=-132,-118
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/rf_fetch -- python3 tools/profile_traffic.py > $out/runfilter_traffic_fetch.json 2> $out/rf_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/rf_write -- python3 tools/profile_traffic.py > $out/runfilter_traffic_write.json 2> $out/rf_write.err
sum=$out/runfilter_pmc_summary_thr$thr.txt
python3 tools/pmc_summary.py $out/rf_fetch $out/rf_write > $sum
rm -rf $out/rf_fetch $out/rf_write
tail -1 $out/runfilter_traffic_fetch.json >> $sum
cat $sum
