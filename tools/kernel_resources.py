#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table from a `hipcc -Rpass-analysis=kernel-resource-usage` log:
tools/kernel_resources.py build_remarks.txt [name-filter]"""
import re, subprocess, sys
rows, cur = [], None
for ln in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark: (?:[^ ]+:\d+:\d+: )? *(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = r["name"]
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>4} spillV {r.get('VGPRs Spill','?'):>3} spillS {r.get('SGPRs Spill','?'):>4} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
