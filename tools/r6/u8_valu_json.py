#!/usr/bin/env python3
"""rocprofv3 --pmc pass over tools/r6/u8_valu.py -> an entry of profiles/pmc_valu_u8.json: vector instructions (wave level) the scan
kernels of ONE steady-state step issue, tied to the uint8 scan kernel's machine code by its sha256.
usage: u8_valu_json.py <pmc-dir> <workload json> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

pmc_dir, wl_json, out = sys.argv[1:4]
wl = json.loads(open(wl_json).read().strip().splitlines()[-1])
rows = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per dispatch (in dispatch order)
for path in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"])):
            if "rt::stft_scan" in r.get("Kernel_Name", ""):
                rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
steps = wl["steps"]
per_step = defaultdict(float)
kernels = {}
for k, c in rows.items():
    # the last half of the steps: AUTO has settled on its level by then (every kernel of that level runs once per step)
    n = len(c["SQ_INSTS_VALU"])
    tail = max(1, min(n, steps // 2))
    kernels[k] = {name: sum(v[-tail:]) / tail for name, v in c.items()}
    kernels[k]["dispatches"] = n
    if n >= steps // 2:
        for name, v in c.items():
            per_step[name] += sum(v[-tail:]) / tail
doc = {}
if os.path.exists(out):
    doc = json.load(open(out))
doc[wl["block"]] = {
    "workload": wl,
    "scan_kernel_sha256": bench.scan_kernel_sha256(symbol=bench.SCAN_KERNEL_SYMBOL_U8),
    "insts_valu_per_step": per_step.get("SQ_INSTS_VALU"),
    "counters_per_step": dict(per_step),
    "kernels": kernels,
    "units": "SQ_INSTS_VALU: wave-level vector instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*: quad-cycles summed over waves; SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs",
    "source": "tools/r6/u8_valu.sh (rocprofv3 --pmc over tools/r6/u8_valu.py)",
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in doc[wl["block"]].items() if k != "kernels"}))
