#!/bin/bash
# Round 3, exploration 5: conditional step below the chunk at nperseg 4096 (instead of the halo segment); nperseg 2048 chunk length
out=gpurun_out/r3h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # tag lib args
  RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/$2 timeout -k 10 300 python bench.py $3 --lanes 1 --steps 12 --warmup 4 --settle 8 --no-cpu-baseline --isolated-steps 0 2>>$out/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['records_per_step'], d['config']['fallbacks'])" >> $out/sweep.txt || exit 1
}
for rep in 1 2; do
run c5_default librt_analyze.so "--workload config5 --total-streams 1024"
run c5_below16 librt_var_below16.so "--workload config5 --total-streams 1024"
done
for L in 0 32 48 62 77; do run "n2048 L=$L" librt_analyze.so "--nperseg 2048 --streams 512 --segs-per-chunk $L"; done
for L in 0 28 31 32 37; do run "c3 L=$L" librt_analyze.so "--workload config3 --segs-per-chunk $L"; done
cat $out/sweep.txt
