#!/bin/bash
# config-2 geometry (and the default geometry) with the noise floor FAR over the absolute threshold (+6 / +10 dB): the
# chunk-bit pre-filter loses its selectivity there, the exact one (SNR-aware bits) keeps it
#   tools/r3/high_floor.sh <tag>
out=gpurun_out/${1:-r3hf}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for floor in -84 -80; do
  for mode in auto prefilter runfilter dense; do
    timeout -k 10 300 python bench.py --noise-dbw $floor --threshold-dbw -90 --mode $mode --steps 40 --warmup 10 --settle 20 --isolated-steps 0 --cpu-streams 16 2>>$out/err.txt | tail -1 >> $out/high_floor.jsonl || echo "{\"failed\": \"$mode $floor\"}" >> $out/high_floor.jsonl
  done
done
for floor in -84 -80; do
  for mode in auto runfilter dense; do
    timeout -k 10 300 python bench.py --sample-rate 300000 --streams 4096 --noise-dbw $floor --mode $mode --steps 20 --warmup 5 --settle 20 --isolated-steps 0 --cpu-streams 16 2>>$out/err.txt | tail -1 >> $out/high_floor.jsonl || echo "{\"failed\": \"$mode $floor\"}" >> $out/high_floor.jsonl
  done
done
python - $out/high_floor.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "failed" in d: print(d); continue
    c = d["config"]; p = d.get("parity") or {}
    print(f"{c['workload']} fs {c.get('sample_rate')} floor {c['noise_floor_dbw']} mode {c['mode']:9s} fallbacks {c['fallbacks']:2d} value {d['value']:9.1f} MS/s  ms/step {d['ms_per_step']:.3f}  records {c['records_per_step']}  cells {c['candidate_cells_per_step']}  parity {p.get('streams_mismatched')}/{p.get('streams_checked')}")
PY
