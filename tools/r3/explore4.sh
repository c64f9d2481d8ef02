#!/bin/bash
# Round 3, exploration 4: finer sweep of segments per chunk (one lane, bench.py)
out=gpurun_out/r3d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # file workload-args L
  timeout -k 10 300 python bench.py $2 --lanes 1 --steps 12 --warmup 4 --settle 8 --no-cpu-baseline --isolated-steps 0 --segs-per-chunk $3 2>>$out/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 L=$3', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['records_per_step'], d['config']['fallbacks'])" >> $out/sweep.txt || exit 1
}
for L in 32 40 52 53 61 66 71 79; do run c5_1024 "--workload config5 --total-streams 1024" $L; done
for L in 32 53 61 71; do run c5_4096 "--workload config5 --total-streams 4096" $L; done
for L in 28 30 31 32 36; do run c3_4096 "--workload config3" $L; done
cat $out/sweep.txt
