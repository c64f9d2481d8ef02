#!/bin/bash
# what AUTO's probes of the level below cost at the reference's default geometry with the noise floor over the threshold:
# runs of 20 / 80 / 200 timed steps (AUTO) next to the explicit exact pre-filter
#   tools/r3/auto_probe_cost.sh <tag>
out=gpurun_out/${1:-r3probe}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--sample-rate 300000 --streams 4096 --warmup 5 --settle 20 --isolated-steps 0 --cpu-streams 16 --noise-dbw -88"
for steps in 20 80 200; do
  timeout -k 10 400 python bench.py $common --steps $steps --mode auto 2>>$out/err.txt | tail -1 >> $out/probe.jsonl || exit 1
done
timeout -k 10 300 python bench.py $common --steps 80 --mode runfilter 2>>$out/err.txt | tail -1 >> $out/probe.jsonl || exit 1
python - $out/probe.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln); c = d["config"]; p = d.get("parity") or {}
    print(f"mode {c['mode']:9s} steps {d['steps']:3d} fallbacks {c['fallbacks']:2d} value {d['value']:9.1f} MS/s  ms/step {d['ms_per_step']:.3f}  parity {p.get('streams_mismatched')}/{p.get('streams_checked')}")
PY
