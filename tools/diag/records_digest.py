"""sha256 of the record arrays a build returns for a fixed set of synthetic batches (all nperseg values, both input
formats, dense / sparse / prefilter, two consecutive buffers): two builds of the kernels that round identically print
identical lines.  usage: RT_ANALYZE_LIB=<.so> python tools/diag/records_digest.py"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyradiotracking_amd import synth  # noqa: E402
from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients  # noqa: E402

for nperseg in (256, 512, 1024, 2048, 4096):
    for window in ("hamming", "blackman"):  # with / without the linearity detrend
        fs, n_streams = 2048000, 6
        blen = nperseg * (40960 // nperseg + 7) + 13
        w = window_coefficients(window, nperseg)
        iq = []
        for s in range(n_streams):
            rng = np.random.default_rng([5, nperseg, s])
            pulses = synth.random_pulses(rng, 2 * blen, fs, w, 14)
            pulses.append(synth.Pulse(blen - int(0.004 * fs), int(0.012 * fs), 1e5 * (s - 3), synth.amp_for_peak_dbw(-70.0, w, fs)))
            iq.append(synth.make_stream(synth.StreamSpec(2 * blen, fs, pulses, dc=complex(1e-3, -2e-3)), seed=50 + s).reshape(2, blen))
        iq = np.stack(iq)
        raw = synth.quantize_u8(iq.reshape(n_streams, -1), gain=5.0).reshape(n_streams, 2, -1)  # (|x| stays under 1: no clipping)
        for mode in ("sparse", "dense"):
            for u8 in (False, True):
                kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=window, signal_min_duration_ms=2.0)
                if u8:
                    kw["signal_threshold_dbw"] = -90.0 + 14.0  # the gain of 5 is 14 dB
                an = BatchSignalAnalyzer([str(i) for i in range(n_streams)], sdr_callback_length=blen, mode=mode, **kw)
                h = hashlib.sha256()
                n = 0
                for k in range(2):
                    if u8:
                        an.enqueue_bytes(np.ascontiguousarray(raw[:, k]))
                    else:
                        an.enqueue(np.ascontiguousarray(iq[:, k]))
                    rec = an.fetch_records()
                    h.update(rec.tobytes())
                    n += len(rec)
                an.close()
                print(nperseg, window, mode, "u8" if u8 else "c64", n, h.hexdigest()[:16], flush=True)
