"""CPU ORACLE for the signal-analysis hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, in NumPy, the algorithm of the reference callback
``SignalAnalyzer.process_samples`` (/root/reference/radiotracking/analyze.py:192-268)
and of the third-party arithmetic it calls.  It exists to *check* the HIP path.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product package ``pyradiotracking_amd`` never
does (it fails loudly when the HIP library is missing).

Parity pin
----------
The reference ships no tests, fixtures or golden vectors for this path
(SURVEY.md section 4), so the pin is the reference itself, imported in the
build container: ``tests/golden/make_golden.py`` drives
``radiotracking.analyze.SignalAnalyzer`` on seeded IQ and stores inputs and
expected records under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this oracle against every stored vector (bit-exact on all fields), and
``tests/test_oracle_vs_reference.py`` re-checks it live against the imported
reference whenever ``/root/reference`` is present.

Third-party arithmetic on the path that is NOT under /root/reference
--------------------------------------------------------------------
``scipy.signal.spectrogram`` (SciPy, unpinned in the reference's
requirements.txt:2; 1.15.3 in this image) and NumPy reductions
(requirements.txt:3; 2.2.6 here).  ``stft_power`` below restates
``scipy/signal/_spectral_py.py`` (``_spectral_helper`` :2066-2155,
``_fft_helper`` :2181-2204, ``_triage_segments`` :2239-2263, constant detrend
``_signaltools.py:3926-3927``); the FFT itself is pocketfft through
``scipy.fft.fft`` -- the same routine SciPy calls (:2198-2202).

Dtype rule: everything runs in the dtype of the input.  With
complex64 IQ the spectrogram is float32 and every comparison against the
Python-float thresholds is a float32 comparison under NumPy-2 promotion.
"""
from __future__ import annotations

import datetime as _dt
from collections import namedtuple
from typing import List, Optional, Sequence, Tuple

import numpy as np
import pytz
import scipy.fft
import scipy.signal

#: raw result of the extractor for one run: integer cell indices plus the
#: float fields exactly as the reference computes them (before Signal()).
OracleRecord = namedtuple(
    "OracleRecord",
    "fi start end start_dt duration_s max_dbw avg_dbw std_db noise_dbw snr_db",
)

#: what the reference puts on its queue, plus the integer provenance.
OracleSignal = namedtuple(
    "OracleSignal",
    "device ts frequency duration max avg std noise snr fi start end",
)


def to_db(v):
    """radiotracking/__init__.py:13-17."""
    return 10 * np.log10(v)


def db_to_linear(d):
    """radiotracking/__init__.py:20-22."""
    return 10 ** (d / 10)


# --------------------------------------------------------------------------
# STFT power  (analyze.py:234-241 -> scipy.signal.spectrogram)
# --------------------------------------------------------------------------
def window_coefficients(window, nperseg: int) -> np.ndarray:
    """scipy/_spectral_py.py:2239-2263 (_triage_segments): a string/tuple goes
    through ``get_window`` (periodic, fftbins=True); anything else is taken as
    the coefficient array and must have length ``nperseg``."""
    if isinstance(window, (str, tuple)):
        return scipy.signal.get_window(window, nperseg)
    win = np.asarray(window)
    if win.ndim != 1:
        raise ValueError("window must be 1-D")
    if win.shape[0] != nperseg:
        raise ValueError("value specified for nperseg is different from length of window")
    return win


def stft_power(x: np.ndarray, fs, window, nperseg: int):
    """Two-sided PSD spectrogram with noverlap=0, detrend='constant',
    scaling='density', mode='psd' -- the only configuration the reference uses
    (analyze.py:234-241; every other argument is SciPy's default).

    Returns ``(freqs[F] f64, times[T] f64, S[F, T])`` with ``S`` a transposed
    view of the segment-major array, as SciPy returns it.
    """
    x = np.asarray(x)
    out_dtype = np.result_type(x, np.complex64)  # _spectral_py.py:1981
    win = window_coefficients(window, nperseg)
    if np.result_type(win, np.complex64) != out_dtype:  # :2083-2084
        win = win.astype(out_dtype)
    scale = 1.0 / (fs * (win * win).sum())  # :2086-2087 (density)

    n_seg = x.shape[-1] // nperseg  # noverlap=0, tail dropped (:2185-2188)
    seg = x[: n_seg * nperseg].reshape(n_seg, nperseg)
    seg = seg - np.mean(seg, axis=-1, keepdims=True)  # :2191, _signaltools.py:3926
    seg = win * seg  # :2194
    spec = scipy.fft.fft(seg, n=nperseg)  # :2198-2202, two-sided
    spec = np.conjugate(spec) * spec  # :2126
    spec *= scale  # :2128
    spec = spec.astype(out_dtype).real  # :2141-2145

    freqs = scipy.fft.fftfreq(nperseg, 1 / fs)  # :2113
    times = np.arange(nperseg / 2, x.shape[-1] - nperseg / 2 + 1, nperseg) / float(fs)  # :2136
    return freqs, times, np.moveaxis(spec, -1, 0)  # :2153


# --------------------------------------------------------------------------
# Run extraction  (analyze.py:330-452)
# --------------------------------------------------------------------------
class ExtractParams:
    """Derived analyzer parameters (analyze.py:101-117)."""

    def __init__(
        self,
        signal_threshold_dbw: float = -90.0,
        snr_threshold_db: float = 5.0,
        signal_min_duration_ms: float = 8,
        signal_max_duration_ms: float = 40,
        calibration_db: float = 0.0,
    ):
        self.calibration_db = calibration_db
        self.signal_min_duration = signal_min_duration_ms / 1000  # :113
        self.signal_max_duration = signal_max_duration_ms / 1000  # :114
        self.signal_threshold = db_to_linear(signal_threshold_dbw + calibration_db)  # :115
        self.snr_threshold = db_to_linear(snr_threshold_db)  # :116


def extract_records(
    times: np.ndarray,
    spec: np.ndarray,
    spec_last: Optional[np.ndarray],
    p: ExtractParams,
) -> List[OracleRecord]:
    """Strided-probe plateau extraction, following analyze.py:349-452 step by
    step (same comparisons, same NumPy scalar types, same order of results:
    frequency bin ascending, then time)."""
    found: List[OracleRecord] = []
    n_t = len(times)
    if n_t == 0:  # :351-352
        return found

    # :354 -- raises IndexError for n_t == 1, like the reference
    min_cells = p.signal_min_duration / (times[1] - times[0])
    step = max(1, int(min_cells))  # :364
    thr = p.signal_threshold
    snr_thr = p.snr_threshold

    for fi, row in enumerate(spec):  # :357
        row_mean = None  # lazy (:359, :374-375)
        resume_at = 0  # :361
        for ti in range(0, len(row), step):  # :364
            if ti < resume_at:  # :366-367
                continue
            if row[ti] < thr:  # :370
                continue
            if row_mean is None:
                row_mean = np.mean(row)  # :375
            if row[ti] / row_mean < snr_thr:  # :378
                continue

            # walk down (:382-398); negative indices address the previous buffer
            lo = ti
            lo_limit = 0 if spec_last is None else -len(spec_last[0]) + 1  # :383
            while lo > lo_limit:
                cell = spec_last[fi, lo] if lo < 0 else row[lo]  # :385-388
                if cell < thr:  # :391
                    break
                if cell / row_mean < snr_thr:  # :395
                    break
                lo -= 1

            # walk up (:401-412)
            hi = ti
            while hi < len(row):
                if row[hi] < thr:
                    resume_at = hi
                    break
                if row[hi] / row_mean < snr_thr:
                    resume_at = hi
                    break
                hi += 1

            if hi == len(row):  # :415-417  run laps into the next buffer
                continue

            hi_dt = times[hi]  # :420
            lo_dt = -times[-lo] if lo < 0 else times[lo]  # :422-425
            duration_s = hi_dt - lo_dt  # :427
            if duration_s < p.signal_min_duration:  # :429
                continue
            if duration_s > p.signal_max_duration:  # :431
                continue

            if lo < 0:  # :437-440
                cells = np.concatenate((spec_last[fi][lo:], row[:hi]))
            else:
                cells = row[lo:hi]

            cell_mean = np.mean(cells)  # :443
            found.append(
                OracleRecord(
                    fi=fi,
                    start=lo,
                    end=hi,
                    start_dt=lo_dt,
                    duration_s=duration_s,
                    max_dbw=to_db(np.max(cells)) - p.calibration_db,  # :442
                    avg_dbw=to_db(cell_mean) - p.calibration_db,  # :444
                    std_db=np.std(to_db(cells)),  # :445
                    noise_dbw=to_db(row_mean),  # :446
                    snr_db=to_db(cell_mean / row_mean),  # :447
                )
            )
    return found


def records_to_signals(
    records: Sequence[OracleRecord],
    freqs: np.ndarray,
    ts_start: _dt.datetime,
    device: str,
    center_freq: float,
) -> List[OracleSignal]:
    """Field construction of analyze.py:360, 428, 434, 449 together with the
    coercions of ``Signal.__init__`` (radiotracking/__init__.py:150-170)."""
    out = []
    for r in records:
        ts = ts_start + _dt.timedelta(seconds=r.start_dt)  # :434
        out.append(
            OracleSignal(
                device=device,
                ts=ts.astimezone(pytz.utc),  # :449
                frequency=float(freqs[r.fi] + center_freq),  # :360
                duration=_dt.timedelta(seconds=r.duration_s),  # :428
                max=float(r.max_dbw),
                avg=float(r.avg_dbw),
                std=float(r.std_db),
                noise=float(r.noise_dbw),
                snr=float(r.snr_db),
                fi=r.fi,
                start=r.start,
                end=r.end,
            )
        )
    return out


# --------------------------------------------------------------------------
# Shadow filter  (analyze.py:282-328)
# --------------------------------------------------------------------------
def shadow_index(sig, others) -> Optional[int]:
    """analyze.py:283-313: index of the first entry of ``others`` that overlaps
    ``sig`` in time (inclusive bounds) and is strictly louder, else None."""
    for i, other in enumerate(others):
        if sig.ts > other.ts + other.duration:  # :302
            continue
        if sig.ts + sig.duration < other.ts:  # :306
            continue
        if other.max > sig.max:  # :310
            return i
    return None


def filter_shadows(signals):
    """analyze.py:315-328: compare every signal against the *unfiltered* list,
    keep those that are nobody's shadow, preserve order."""
    verdict = [shadow_index(s, signals) for s in signals]
    return [s for s, v in zip(signals, verdict) if v is None]


# --------------------------------------------------------------------------
# The callback  (analyze.py:192-268, minus SDR/clock/state plumbing)
# --------------------------------------------------------------------------
class OracleAnalyzer:
    """One stream's analyzer state: parameters + the previous spectrogram.

    ``process(buffer, ts_start)`` performs analyze.py:231-268 for one buffer
    and returns ``(all_signals, kept_signals)``.
    """

    def __init__(
        self,
        device: str = "0",
        calibration_db: float = 0.0,
        sample_rate: int = 300000,
        center_freq: int = 150150000,
        fft_nperseg: int = 256,
        fft_window="hamming",
        signal_min_duration_ms: float = 8,
        signal_max_duration_ms: float = 40,
        signal_threshold_dbw: float = -90.0,
        snr_threshold_db: float = 5.0,
        **_ignored,
    ):
        self.device = device
        self.sample_rate = sample_rate
        self.center_freq = center_freq
        self.fft_nperseg = fft_nperseg
        self.fft_window = fft_window
        self.params = ExtractParams(
            signal_threshold_dbw,
            snr_threshold_db,
            signal_min_duration_ms,
            signal_max_duration_ms,
            calibration_db,
        )
        self.spec_last: Optional[np.ndarray] = None  # analyze.py:128

    def reset(self):
        self.spec_last = None

    def process(self, buffer: np.ndarray, ts_start: _dt.datetime):
        freqs, times, spec = stft_power(buffer, self.sample_rate, self.fft_window, self.fft_nperseg)
        records = extract_records(times, spec, self.spec_last, self.params)  # :245
        signals = records_to_signals(records, freqs, ts_start, self.device, self.center_freq)
        kept = filter_shadows(signals)  # :248
        self.spec_last = spec  # :268
        return signals, kept


def analyze_stream(buffers, ts_starts, **kwargs) -> Tuple[list, list]:
    """Convenience: run consecutive buffers of one stream through a fresh
    ``OracleAnalyzer``; returns per-buffer lists (all, kept)."""
    oa = OracleAnalyzer(**kwargs)
    every, kept = [], []
    for buf, ts in zip(buffers, ts_starts):
        a, k = oa.process(buf, ts)
        every.append(a)
        kept.append(k)
    return every, kept
