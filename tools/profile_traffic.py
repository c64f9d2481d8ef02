#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic of the scan kernel).

Runs, on config 2 (256 streams x 2.048 MS, nperseg 256):
  * CAL launches of the load-only calibration kernel (stft_scan<1,3>), whose
    byte count is known exactly, and
  * STEPS full analysis steps (stft_scan<1,0> + detect_sparse).
Prints the exact byte counts as JSON so tools/pmc_summary.py can turn the
FETCH_SIZE / WRITE_SIZE rows into corrected bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/profile_traffic.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/profile_traffic.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    # torch-free on purpose: rocprofv3 --pmc (ROCm 7.2) crashes at start-up when torch's
    # bundled ROCm 7.0 runtime is loaded; the library itself only needs /opt/rocm's HIP.
    os.environ["RT_NO_TORCH"] = "1"
    import numpy as np

    from pyradiotracking_amd import _native, synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

    S = int(os.environ.get("RT_PROF_STREAMS", "256"))
    fs, nperseg = int(os.environ.get("RT_PROF_FS", "2048000")), int(os.environ.get("RT_PROF_NPERSEG", "256"))
    blen = int(os.environ.get("RT_PROF_SAMPLES", str(fs)))
    cal = int(os.environ.get("RT_PROF_CAL", "3"))
    steps = int(os.environ.get("RT_PROF_STEPS", "3"))
    win = window_coefficients("hamming", nperseg)
    # 8 distinct streams (noise + 4-8 pulses each), tiled to S: traffic does not depend on content
    base = []
    for s in range(8):
        rng = np.random.default_rng([1000, s])
        k = int(rng.integers(4, 9))
        peak = os.environ.get("RT_PROF_PEAK_DBW")  # "lo,hi": pulse peak powers (default: synth's -80 .. -60 dBW)
        extra = dict(peak_dbw=tuple(float(x) for x in peak.split(","))) if peak else {}
        pulses = synth.random_pulses(rng, blen, fs, win, k, keep_clear_tail=2 * nperseg, **extra)
        base.append(synth.make_stream(synth.StreamSpec(blen, fs, pulses), 1000 + s))
    dev = _native.DeviceBuffer(0, S * blen * 8)
    for s in range(S):
        _native.load_library().rt_dev_upload(0, dev.ptr + s * blen * 8, base[s % 8].ctypes.data, blen * 8)
    an = BatchSignalAnalyzer([str(i) for i in range(S)], sdr_callback_length=blen, sample_rate=fs, fft_nperseg=nperseg, mode=os.environ.get("RT_PROF_MODE", "sparse"),
                             signal_threshold_dbw=float(os.environ.get("RT_PROF_THRESHOLD_DBW", "-90")))
    for _ in range(cal):
        an.native.calibrate_read(dev.ptr, blen, blen)
    n_hot = n_rec = 0
    for _ in range(steps):
        an.enqueue(dev.ptr, n_samples=blen)
        rec = an.fetch_records()
        n_hot, n_rec = an.native.call_info().n_hot, len(rec)
    T = blen // nperseg
    L = int(an.native.call_info().segs_per_chunk)  # the handle's own choice (rt_call_info): part of what the traffic figure is keyed on
    print(json.dumps({
        "streams": S, "segments": T, "nperseg": nperseg, "segs_per_chunk": L,
        "algorithmic_bytes": S * T * nperseg * 8,
        "scan_read_bytes_exact": S * T * nperseg * 8,  # the load-only calibration launch reads every segment once
        "candidate_cells": int(n_hot), "records": int(n_rec),
    }))


if __name__ == "__main__":
    main()
