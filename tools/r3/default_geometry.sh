#!/bin/bash
# the reference's default geometry (300 kS/s, nperseg 256, 8 - 40 ms, -90 dBW; __main__.py:48-64) with the noise floor
# around the threshold: RT_MODE_AUTO (sparse -> exact run-length pre-filter -> dense) against the dense path
#   tools/r3/default_geometry.sh <tag>
out=gpurun_out/${1:-r3dg}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--sample-rate 300000 --streams 4096 --steps 20 --warmup 5 --settle 20 --isolated-steps 0 --cpu-streams 64"
for floor in -92 -90 -88 -86; do
  for mode in auto runfilter dense; do
    timeout -k 10 300 python bench.py $common --noise-dbw $floor --mode $mode 2>>$out/err.txt | tail -1 >> $out/floors.jsonl || exit 1
  done
done
timeout -k 10 300 python bench.py $common 2>>$out/err.txt | tail -1 >> $out/floors.jsonl || exit 1
python - $out/floors.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln); c = d["config"]; p = d.get("parity") or {}
    print(f"floor {c['noise_floor_dbw']} mode {c['mode']:9s} fallbacks {c['fallbacks']:2d} value {d['value']:9.1f} MS/s  ms/step {d['ms_per_step']:.3f}  records {c['records_per_step']}  cells {c['candidate_cells_per_step']}  parity {p.get('streams_mismatched')}/{p.get('streams_checked')} worst {p.get('worst_db_difference')}")
PY
