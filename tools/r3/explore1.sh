#!/bin/bash
# Round 3, first exploration call on the GPU box:
#   1. tools/micro/valu_rate.hip (scalar and packed f32 issue rates at 1..4 waves per SIMD)
#   2. stage ablation of the scan kernel (RT_ABLATE builds made by tools/ablate.sh, -DRT_EXP_ALIAS=63 by tools/variant.sh alias)
#      at nperseg 1024 and 4096
#   3. one-lane bench lines of configs 2, 3 and the 1 024-stream share of config 5 (this box's baseline)
# Every step stops the script when it fails or times out.
out=gpurun_out/r3a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o /tmp/valu_rate tools/micro/valu_rate.hip 2>/dev/null || exit 1
timeout -k 10 120 /tmp/valu_rate > $out/valu_rate.txt 2>&1 || exit 1
echo "micro done"
for g in "4096 3200000 512" "1024 2400000 1024"; do
  for v in analyze ablate_1 ablate_2 ablate_3 ablate_4 ablate_5 ablate_6 ablate_8 ablate_7 ablate_9 var_alias; do
    RT_ANALYZE_LIB=$PWD/pyradiotracking_amd/librt_$v.so timeout -k 10 200 python tools/ablate_large.py $g 2>>$out/ablate.err | tail -1 >> $out/ablate.txt || exit 1
  done
  echo "ablation $g done"
done
for w in "config5 --total-streams 1024" "config3" "config2"; do
  timeout -k 10 300 python bench.py --workload $w --lanes 1 --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 0 2>>$out/bench.err | tail -1 >> $out/bench_one_lane.jsonl || exit 1
done
echo "bench done"
