#!/usr/bin/env python3
"""Stage timing of the scan kernel for one nperseg: load-only (rt_calibrate_read) and the full step, HIP-event
timed by the library (RT_FLAG_TIMING).  Run once per RT_ABLATE build (RT_ANALYZE_LIB=...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyradiotracking_amd import synth
from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

nperseg = int(sys.argv[1]); fs = float(sys.argv[2]); S = int(sys.argv[3])
blen = int(fs)
win = window_coefficients("hamming", nperseg)
iq = synth.make_batch_device(S, blen, fs, win, seed=1, device="cuda:0")
an = BatchSignalAnalyzer([str(i) for i in range(S)], sdr_callback_length=blen, sample_rate=fs, fft_nperseg=nperseg,
                         mode="sparse", timing=True, signal_max_duration_ms=float(os.environ.get("RT_MAXDUR_MS", "40")), hip_stream=torch.cuda.current_stream().cuda_stream)
T = blen // nperseg
gb = S * T * nperseg * 8 / 1e9
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ms_load = timed(lambda: an.native.calibrate_read(iq.data_ptr(), blen, blen))
acc = 0.0
for i in range(12):
    an.enqueue(iq); an.fetch_records()
    if i >= 2: acc += an.native.call_info().ms_stft
ms = acc / 10
print(f"N={nperseg} S={S} lib={os.environ.get('RT_ANALYZE_LIB','default').split('/')[-1]} load-only {ms_load:.3f} ms ({gb/ms_load:.0f} GB/s)  scan {ms:.3f} ms ({gb/ms:.0f} GB/s)")
an.close()
