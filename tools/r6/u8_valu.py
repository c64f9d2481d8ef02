#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc pass that prices the uint8 wire-format path against the roofline that bounds it: vector-
instruction issue, not HBM (2 bytes per sample).  Torch-free (rocprofv3 --pmc and torch's bundled runtime do not mix).
RT_PROF_BLOCK = config2_uint8 (256 streams x 2.048 MS, threshold -80 dBW: the sparse scan) | default_geometry_uint8_noise_floor
(4 096 streams x 300 kS/s, threshold -91 dBW under the quantisation noise: AUTO -> exact run-length pre-filter) -- the two uint8
blocks of bench.py's other_configs, same geometry, noise and pulse levels (eight distinct streams tiled over the batch)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    os.environ["RT_NO_TORCH"] = "1"
    import numpy as np

    from pyradiotracking_amd import _native, synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

    block = os.environ.get("RT_PROF_BLOCK", "config2_uint8")
    if block == "config2_uint8":
        S, fs, thr = 256, 2048000, -80.0
    else:
        S, fs, thr = 4096, 300000, -91.0
    nperseg, blen = 256, fs
    steps = int(os.environ.get("RT_PROF_STEPS", "6"))
    win = window_coefficients("hamming", nperseg)
    base = []
    for s in range(8):
        rng = np.random.default_rng([1000, s])
        pulses = synth.random_pulses(rng, blen, fs, win, int(rng.integers(4, 9)), keep_clear_tail=2 * nperseg, peak_dbw=(-62.0, -48.0))
        base.append(np.ascontiguousarray(synth.quantize_u8(synth.make_stream(synth.StreamSpec(blen, fs, pulses, noise_sigma=0.012), 1000 + s))))
    dev = _native.DeviceBuffer(0, S * blen * 2)
    lib = _native.load_library()
    for s in range(S):
        lib.rt_dev_upload(0, dev.ptr + s * blen * 2, base[s % 8].ctypes.data, blen * 2)
    an = BatchSignalAnalyzer([str(i) for i in range(S)], sdr_callback_length=blen, sample_rate=fs, fft_nperseg=nperseg, mode="auto", signal_threshold_dbw=thr)
    info = None
    for _ in range(steps):
        an.enqueue_bytes(dev.ptr, n_samples=blen, stream_stride=blen)
        rec = an.fetch_records()
        info = an.native.call_info()
    print(json.dumps({"block": block, "streams": S, "segments": blen // nperseg, "nperseg": nperseg, "steps": steps, "segs_per_chunk": int(info.segs_per_chunk),
                      "mode_used": int(info.mode_used), "records": int(len(rec)), "candidate_cells": int(info.n_hot), "samples_per_step": S * (blen // nperseg) * nperseg}))


if __name__ == "__main__":
    main()
