"""Occupancy over time from an RT_STAMPS_DUMP file: python tools/r3/timeline.py dump.bin"""
import sys
import numpy as np
k = 16
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, k)
a = a[a[:, 11] > 0]
start, end = a[:, 14].astype(np.int64), a[:, 15].astype(np.int64)
t0 = start.min()
start -= t0; end -= t0
span = end.max()
print(f"{len(a)} waves, span {span/100:.1f} us, wave life mean {np.mean(end-start)/100:.1f} us (min {np.min(end-start)/100:.1f}, max {np.max(end-start)/100:.1f})")
edges = np.linspace(0, span, 41)
for lo, hi in zip(edges[:-1], edges[1:]):
    # average number of resident waves in the bin
    ov = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum() / (hi - lo)
    starts = int(((start >= lo) & (start < hi)).sum())
    print(f"{lo/100:8.1f} us: {ov/1024:5.2f} waves/SIMD   {starts:6d} wave starts  " + "#" * int(ov / 1024 * 20))
# first-start spread and gaps: per workgroup order
order = np.argsort(start)
print("wave starts: first", start[order[0]] / 100, "us; 10%", np.percentile(start, 10) / 100, "50%", np.percentile(start, 50) / 100)
