#!/usr/bin/env python3
"""Randomised soak of extract_signals + filter_shadow_signals on caller-supplied power maps (rt_extract ->
detect_dense) against the oracle: random numbers of bins and columns, plateaus planted on every edge class
(start / end of the map, look-back into a previous map of any length incl. 1, exact-threshold cells, zero and
NaN cells), random thresholds and durations.  Exact powers are planted (a few distinct levels), so decisions do
not sit on round-off: every record must match.  usage: soak_extract.py [seconds] [seed]"""
import datetime
import os
import sys
import time

import numpy as np
import scipy.fft

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import analyze_oracle as oracle  # noqa: E402
from pyradiotracking_amd.analyze import SignalAnalyzer  # noqa: E402

TS0 = datetime.datetime(2024, 1, 1)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
n_cases = n_rec = n_bad = n_ref_raises = 0
case = 0
cache = {}
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed0, case])
    nperseg = int(rng.choice([256, 1024]))
    fs = int(rng.choice([300000, 2048000]))
    hop = nperseg / fs
    min_ms = float(rng.choice([0.0, 2 * hop * 1e3, 5 * hop * 1e3]))
    max_ms = float(min_ms + rng.choice([6, 20, 60]) * hop * 1e3)
    thr_db, snr_db = float(rng.choice([-90.0, -80.0])), float(rng.choice([0.0, 5.0]))
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, signal_min_duration_ms=min_ms, signal_max_duration_ms=max_ms,
              signal_threshold_dbw=thr_db, snr_threshold_db=snr_db)
    key = tuple(sorted(kw.items()))
    if key not in cache:
        if len(cache) > 6:
            cache.clear()
        cache[key] = SignalAnalyzer("0", sdr_callback_length=4096, **kw)
    an = cache[key]
    F = int(rng.choice([1, 3, 16, 100, 256, 700]))
    T = int(rng.integers(2, 200))
    thr = np.float32(10 ** (thr_db / 10))
    floor = np.float32(thr * 10 ** (-rng.uniform(0.5, 3)))
    def noise_map(t):
        m = rng.exponential(1.0, (F, t)).astype(np.float32) * floor
        return m
    cur = noise_map(T)
    has_last = rng.random() < 0.7
    T_last = int(rng.choice([1, 2, 3, 50, 300])) if has_last else 0
    last = noise_map(T_last) if has_last else None
    # plant plateaus: levels well clear of both thresholds (x4 .. x1000 of thr and far above snr * row mean)
    for _ in range(int(rng.integers(0, 12))):
        fi = int(rng.integers(0, F))
        ln = int(rng.integers(1, 70))
        kind = rng.integers(0, 5)
        level = np.float32(thr * rng.choice([4.0, 32.0, 1000.0]))
        if kind == 0:      # inside
            st = int(rng.integers(0, max(1, T - ln)))
            cur[fi, st:st + ln] = level
        elif kind == 1:    # touching the end of the map
            cur[fi, max(0, T - ln):] = level
        elif kind == 2:    # from the start, with or without continuation in the previous map
            cur[fi, :min(T, ln)] = level
            if has_last and rng.random() < 0.7:
                back = int(rng.integers(1, T_last + 1))
                last[fi, T_last - back:] = level
        elif kind == 3:    # exact threshold / zero / NaN cells inside a plateau
            st = int(rng.integers(0, max(1, T - ln)))
            cur[fi, st:st + ln] = level
            cur[fi, min(T - 1, st + ln // 2)] = rng.choice([thr, np.float32(0.0), np.float32(np.nan), np.nextafter(thr, np.float32(0))])
        else:              # two plateaus in neighbouring bins overlapping in time (shadow filter)
            st = int(rng.integers(0, max(1, T - ln)))
            cur[fi, st:st + ln] = level
            cur[(fi + 1) % F, st + ln // 3:st + ln // 3 + ln] = np.float32(level * rng.choice([0.5, 1.0, 2.0]))
    freqs = scipy.fft.fftfreq(F, 1 / fs) if F > 1 else np.array([0.0])
    times = (nperseg / 2 + np.arange(T) * nperseg) / float(fs)
    an._spectrogram_last = last
    try:
        sigs = an.extract_signals(freqs, times, cur, TS0)
    except Exception as e:
        print(f"case {case}: extract failed: {e}")
        continue
    rec = an._last_records
    p = oracle.ExtractParams(thr_db, snr_db, min_ms, max_ms, 0.0)
    try:
        with np.errstate(all="ignore"):
            want = oracle.extract_records(times, cur, last, p)
            wsig = oracle.records_to_signals(want, freqs, TS0, "0", 150150000)
            kept = oracle.filter_shadows(wsig)
    except IndexError:
        n_ref_raises += 1  # a look-back longer than the current map has columns: the reference raises (analyze.py:422-423)
        continue
    got = [(int(r["fi"]), int(r["start"]), int(r["end"])) for r in rec]
    exp = [(x.fi, x.start, x.end) for x in want]
    kept_ids = {id(x) for x in kept}
    ok = got == exp and [bool(r["shadowed"]) for r in rec] == [id(x) not in kept_ids for x in wsig]
    if ok:
        for g, x in zip(sigs, wsig):
            for name in ("max", "avg", "noise", "snr", "std"):
                a_, b_ = getattr(g, name), getattr(x, name)
                ok = ok and (abs(a_ - b_) < 0.01 or (np.isnan(a_) and np.isnan(b_)) or (np.isinf(a_) and a_ == b_))
            ok = ok and g.ts == x.ts and g.duration == x.duration
    n_cases += 1
    n_rec += len(exp)
    if not ok:
        n_bad += 1
        print(f"MISMATCH case {case}: F={F} T={T} T_last={T_last if has_last else None} min/max={min_ms:.3f}/{max_ms:.3f} thr={thr_db} snr={snr_db}: "
              f"{len(got)} vs {len(exp)} records; extra {sorted(set(got) - set(exp))[:4]} missing {sorted(set(exp) - set(got))[:4]}", flush=True)
print(f"SOAK EXTRACT: {n_cases} cases, {n_rec} oracle records, {n_bad} mismatching cases ({n_ref_raises} more cases skipped: the reference raises IndexError)")
