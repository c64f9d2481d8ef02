#!/bin/bash
# the driver's command, twice (plain), output kept: tools/r5/bench_final.sh <tag>
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
( time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_$i.json 2> $out/bench_$i.err
python3 - $out/bench_$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r=d["roofline"]
print("headline", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "kernel_ms", r["kernel_ms"], "whole", r["whole_path_frac"], "traffic", r["traffic"], "latency", d.get("single_stream_latency_ms"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("config1_single_core_ms_per_s"), "parity", d["parity"]["streams_mismatched"])
for o in d.get("other_configs", []): print("  ", o["name"], o.get("value"), "ms/step", o.get("ms_per_step"), "kernel_ms", o.get("kernel_ms"), "frac", o.get("frac"), "whole", o.get("whole_path_frac"), "mode", o.get("mode"), "parity_bad", o.get("parity_streams_mismatched"), "wall", o.get("wall_s"), o.get("failed") or o.get("skipped") or "")
PY
grep real $out/bench_$i.err
done
