#!/bin/bash
# Config 2 with a noise floor around the reference's default -90 dBW threshold (AUTO mode settles on the run-length
# pre-filter), and the dense path on the same input for comparison: tools/run_noise_regimes.sh out.jsonl
out=${1:-gpurun_out/noise_regimes.jsonl}
: > $out
for n in -92 -90 -88 -86; do
  python3 bench.py --noise-dbw $n --threshold-dbw -90 --steps 60 --warmup 10 --no-cpu-baseline --isolated-steps 0 | tail -1 >> $out || exit 1
done
python3 bench.py --noise-dbw -88 --threshold-dbw -90 --mode dense --steps 60 --warmup 10 --no-cpu-baseline --isolated-steps 0 | tail -1 >> $out || exit 1
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]
    print(d["value"], d["ms_per_step"], c["mode"], c.get("noise_floor_dbw"), c["candidate_cells_per_step"], c["records_per_step"], c["fallbacks"], d.get("parity", {}).get("streams_mismatched"))
PY
