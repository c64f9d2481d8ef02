#!/bin/bash
# uint8 wire format at config-2 geometry with the threshold around the quantisation-noise floor (AUTO settles on the
# run-length pre-filter), and the dense path on the same input: tools/run_u8_noise_regimes.sh out.jsonl
out=${1:-gpurun_out/u8_noise_regimes.jsonl}
: > $out
for t in -80 -97 -98.5 -100; do
  python3 bench.py --input u8 --threshold-dbw $t --steps 60 --warmup 10 --no-cpu-baseline --isolated-steps 0 | tail -1 >> $out || exit 1
done
python3 bench.py --input u8 --threshold-dbw -100 --mode dense --steps 60 --warmup 10 --no-cpu-baseline --isolated-steps 0 | tail -1 >> $out || exit 1
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]
    print(d["value"], d["ms_per_step"], c["mode"], c.get("threshold_dbw"), c["candidate_cells_per_step"], c["records_per_step"], c["fallbacks"])
PY
