#!/usr/bin/env python3
"""Round-off of the scan's transform under a strong tone: the GPU spectrogram (rt_spectrogram_device) and SciPy's float32
spectrogram, each against a float64 transform of the same complex64 samples.  Cells are binned by how far they lie under the
strongest bin of their segment; printed: rms and worst |dB error| per bin of 10 dB.
usage: RT_ANALYZE_LIB=<lib> python tests/perf/fft_round_off.py [nperseg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # the repo root
import numpy as np
from oracle import analyze_oracle as oracle
from pyradiotracking_amd import _native, synth
from pyradiotracking_amd.analyze import BatchSignalAnalyzer

nperseg = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fs, n_seg, S = 3200000, 48, 4
n = n_seg * nperseg
rng = np.random.default_rng(7)
w = oracle.window_coefficients("hamming", nperseg)
iq = np.stack([synth.make_stream(synth.StreamSpec(n, fs, synth.random_pulses(rng, n, fs, w, 6, dur_ms=(4, 30))), 300 + s) for s in range(S)])
b = BatchSignalAnalyzer([str(s) for s in range(S)], sdr_callback_length=n, gpu=0, sample_rate=fs, fft_nperseg=nperseg, fft_window="hamming", mode="dense")
d_iq = _native.DeviceBuffer(0, iq.nbytes); d_iq.upload(iq)
d_out = _native.DeviceBuffer(0, S * n_seg * nperseg * 4)
b.native.spectrogram_device(d_iq.ptr, n, n, d_out.ptr)
got = d_out.download(np.float32, S * n_seg * nperseg).reshape(S, n_seg, nperseg).astype(np.float64)
# float64 reference on the same float32 samples, the float32 window SciPy uses (wc = w.astype(complex64)), float64 arithmetic
wc = w.astype(np.float32).astype(np.float64)
scale = 1.0 / (fs * (wc * wc).sum())
ref = np.empty_like(got); sci = np.empty_like(got)
for s in range(S):
    seg = iq[s].astype(np.complex128).reshape(n_seg, nperseg)
    seg = seg - seg.mean(axis=1, keepdims=True)
    X = np.fft.fft(seg * wc, axis=1)
    ref[s] = (X.real ** 2 + X.imag ** 2) * scale
    sci[s] = oracle.stft_power(iq[s], fs, "hamming", nperseg)[2].T
lvl = 10 * np.log10(ref / ref.max(axis=2, keepdims=True))
hot = 10 * np.log10(ref.max(axis=2, keepdims=True)) > -100  # segments that hold a tone
print(f"lib {os.environ.get('RT_ANALYZE_LIB', 'default')}  nperseg {nperseg}: |dB error| against a float64 transform, cells of segments with a tone, by level under the segment's strongest bin")
for name, x in (("gpu", got), ("scipy-f32", sci)):
    err = np.abs(10 * np.log10(np.maximum(x, 1e-300) / ref))
    row = []
    for lo in range(0, 100, 10):
        m = hot & (lvl <= -lo) & (lvl > -lo - 10)
        row.append(f"{-lo:4d}..{-lo - 10:4d} dB: rms {np.sqrt((err[m] ** 2).mean()):.2e} max {err[m].max():.2e}" if m.any() else f"{-lo:4d}: -")
    print(f"  {name:9s} " + " | ".join(row))
