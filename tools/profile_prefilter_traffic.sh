#!/bin/bash
# HBM traffic of the run-length pre-filter level (flags-only scan + plan + selective pass) on config-2 geometry with the
# noise floor AT the threshold (-160 dBW with the synthetic sigma; pulses 20 .. 34 dB over it): two --pmc passes.
# usage (GPU box): tools/profile_prefilter_traffic.sh <tag>  -> gpurun_out/prof_<tag>/prefilter_pmc_summary.txt
tag=${1:-pf}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export RT_PROF_MODE=prefilter RT_PROF_THRESHOLD_DBW=-160 RT_PROF_PEAK_DBW=-140,-126
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf_fetch -- python3 tools/profile_traffic.py > $out/prefilter_traffic_fetch.json 2> $out/pf_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pf_write -- python3 tools/profile_traffic.py > $out/prefilter_traffic_write.json 2> $out/pf_write.err
python3 tools/pmc_summary.py $out/pf_fetch $out/pf_write > $out/prefilter_pmc_summary.txt
rm -rf $out/pf_fetch $out/pf_write
cat $out/prefilter_pmc_summary.txt; tail -1 $out/prefilter_traffic_fetch.json
