#!/bin/bash
# Round 3, exploration 3: segments per chunk at nperseg 4096 / 1024 (one lane, bench.py)
out=gpurun_out/r3c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 0 48 56 64 71 96; do
  timeout -k 10 300 python bench.py --workload config5 --total-streams 1024 --lanes 1 --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 0 --segs-per-chunk $L 2>>$out/bench.err | tail -1 >> $out/c5_L.jsonl || exit 1
done
for L in 0 48 64 96; do
  timeout -k 10 300 python bench.py --workload config3 --total-streams 2048 --lanes 1 --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 0 --segs-per-chunk $L 2>>$out/bench.err | tail -1 >> $out/c3_L.jsonl || exit 1
done
python - <<'PY'
import json
for f in ("gpurun_out/r3c/c5_L.jsonl", "gpurun_out/r3c/c3_L.jsonl"):
    for ln in open(f):
        d = json.loads(ln)
        print(f.split('/')[-1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["records_per_step"], d["config"]["fallbacks"])
PY
