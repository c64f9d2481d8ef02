#!/usr/bin/env python3
"""Steady-state picture from a rocprofv3 --kernel-trace csv: for the rt:: kernels of the last N steps, per kernel name the
count, mean duration and the share of wall time the GPU ran NOTHING of ours, ONE kernel, or SEVERAL at once; then one step
listed kernel by kernel (start, end relative to the step's first kernel, queue)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rt::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "").replace("rt::", "")[:34]
# the last 40 % of the launches = steady state
k0 = int(len(rows) * 0.6)
sel = rows[k0:]
t0, t1 = sel[0]["s"], max(r["e"] for r in sel)
ev = []
for r in sel:
    ev.append((r["s"], 1)); ev.append((r["e"], -1))
ev.sort()
busy = collections.Counter(); depth = 0; last = t0
for t, d in ev:
    busy[min(depth, 2)] += t - last; last = t; depth += d
tot = t1 - t0
scans = [r for r in sel if "stft_scan" in r["Kernel_Name"]]
print(f"window {tot/1e6:.2f} ms, {len(scans)} scan launches; idle {100*busy[0]/tot:.1f} %  one kernel {100*busy[1]/tot:.1f} %  overlapped {100*busy[2]/tot:.1f} %")
by = collections.defaultdict(list)
for r in sel: by[short(r["Kernel_Name"])].append(r["e"] - r["s"])
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:36s} n={len(v):4d} mean {sum(v)/len(v)/1e3:9.1f} us  sum/window {100*sum(v)/tot:5.1f} %")
# one step, kernel by kernel: from the 4th-last MODE-6 (or MODE-0) scan start
firsts = [r for r in sel if "stft_scan<1, 6" in r["Kernel_Name"] or "stft_scan<1, 0" in r["Kernel_Name"]]
if len(firsts) >= 6:
    a, b = firsts[-6]["s"], firsts[-2]["s"]
    print(f"-- kernels starting in [{0}, {(b-a)/1e3:.0f}] us (4 first-scan launches)")
    for r in sel:
        if a <= r["s"] < b:
            print(f"   {(r['s']-a)/1e3:9.1f} .. {(r['e']-a)/1e3:9.1f}  ({(r['e']-r['s'])/1e3:8.1f})  q{r.get('Queue_Id','?'):>3s}  {short(r['Kernel_Name'])}")
