#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite on the round's first sources, then the bare bench line (with other_configs)
out=gpurun_out/r5a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > $out/tests.txt 2>&1
echo "tests rc=$?" | tee -a $out/tests.txt
tail -3 $out/tests.txt
( time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
tail -c 600 $out/bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5a/bench.json") if l.startswith("{")][0])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("single_stream_latency_ms"), d["cpu_baseline"].get("config1_single_core_ms_per_s"))
print(d["config"]["devices"])
for o in d.get("other_configs", []): print({k: o[k] for k in o if k not in ("workload","kernel","kernel_ms_note")})
PY
