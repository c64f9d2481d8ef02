#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the idle gaps between
consecutive kernels of the steady state (usage: timeline.py <kernel_trace.csv> [n_last_steps])."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
scan_idx = [i for i, r in enumerate(rows) if "stft_scan" in r["Kernel_Name"]]
first = scan_idx[-n_last - 1]
last = scan_idx[-1]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for i in range(first, last):
    r, nx = rows[i], rows[i + 1]
    name = r["Kernel_Name"].split("(")[0][-40:]
    dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    gap[name + " -> " + nx["Kernel_Name"].split("(")[0][-30:]].append(int(nx["Start_Timestamp"]) - int(r["End_Timestamp"]))
span = int(rows[last]["Start_Timestamp"]) - int(rows[first]["Start_Timestamp"])
print(f"steps {n_last}: {span / n_last / 1e3:.1f} us per step (scan start to scan start)")
for k, v in dur.items():
    print(f"  kernel {k:42s} n={len(v):3d} avg {sum(v) / len(v) / 1e3:8.1f} us")
for k, v in gap.items():
    print(f"  gap    {k:75s} n={len(v):3d} avg {sum(v) / len(v) / 1e3:8.1f} us")
