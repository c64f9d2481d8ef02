#!/bin/bash
# quick parity subset + stage stamps + one-lane bench lines: tools/r3/check_and_bench.sh <tag> [full]
tag=${1:-r3j}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$2" = full ]; then sel="gpu"; kexp=""; else sel="gpu"; kexp="spectrogram or golden_iq or batch_of_streams or detrend or look_back or fullsize or full_geometry or uint8 or lanes_give"; fi
timeout -k 10 1000 python -m pytest tests -x -q -m "$sel" ${kexp:+-k "$kexp"} > $out/pytest.txt 2>&1; rc=$?
tail -4 $out/pytest.txt
[ $rc -eq 0 ] || exit $rc
if [ -f pyradiotracking_amd/librt_var_stamps.so ]; then bash tools/r3/stamps.sh $tag > /dev/null || exit 1; cat $out/stamps.txt; fi
for w in "config5 --total-streams 1024" "config3" "config2"; do
  timeout -k 10 300 python bench.py --workload $w --lanes 1 --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 0 2>>$out/bench.err | tail -1 >> $out/bench_one_lane.jsonl || exit 1
done
python tools/show_bench.py $out/bench_one_lane.jsonl
