#!/usr/bin/env python3
"""Randomised soak of the native cross-SDR matcher (rt_match_add, host code) against the imported reference
(radiotracking.match.SignalMatcher; build container only -- needs /root/reference) on random streams: equal
time stamps, out-of-order arrivals, zero/large tolerances, unknown devices, replace-if-louder ties, many devices,
ragged batch sizes.  usage: soak_match.py [seconds] [seed]"""
import datetime
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "/root/reference")
for name in ("cbor2", "paho", "paho.mqtt", "paho.mqtt.client"):
    sys.modules.setdefault(name, types.ModuleType(name))
import radiotracking  # noqa: E402
import radiotracking.match as ref_match  # noqa: E402

from pyradiotracking_amd import match as rtm  # noqa: E402

US = datetime.timedelta(microseconds=1)


class Sink:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
n_cases = n_sig = n_groups = n_bad = 0
case = 0
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed0, case])
    n_dev = int(rng.integers(1, 17))
    devices = [str(i) for i in range(n_dev)]
    params = dict(
        matching_timeout_s=float(rng.choice([0.0, 0.05, 0.5, 2.0, rng.uniform(0, 3)])),
        matching_time_diff_s=float(rng.choice([0.0, 0.001, 0.05, rng.uniform(0, 0.2)])),
        matching_bandwidth_hz=float(rng.choice([0.0, 500.0, rng.uniform(0, 5000)])),
        matching_duration_diff_ms=[None, 0.0, 2.0, 5.5, float(rng.uniform(0, 20))][int(rng.integers(0, 5))],
    )
    sink = Sink()
    ref = ref_match.SignalMatcher(device=devices, signal_queue=sink, **params)
    nat = rtm.NativeMatcher(n_dev, params["matching_timeout_s"], params["matching_time_diff_s"], params["matching_bandwidth_hz"],
                            params["matching_duration_diff_ms"])
    n = int(rng.integers(1, 600))
    t = 1_700_000_000_000_000 + int(rng.integers(0, 10**9))
    n_tags = int(rng.integers(1, 10))
    jitter = int(rng.choice([0, 1, 3000, 500000]))
    rec = np.zeros(n, dtype=rtm.SIGNAL_DTYPE)
    for i in range(n):
        t += int(rng.choice([0, 0, 1, rng.integers(0, 200_000), rng.integers(0, 3_000_000)]))
        dev = int(rng.integers(0, n_dev + (1 if rng.random() < 0.1 else 0)))  # sometimes a device the matcher does not know
        freq = 150e6 + float(rng.integers(0, n_tags)) * 1500.0 + float(rng.choice([0.0, rng.uniform(-5, 5), rng.uniform(-600, 600)]))
        dur = int(rng.choice([0, 1, rng.integers(8000, 30000)]))
        avg = float(rng.choice([-60.0, rng.uniform(-80, -40)]))
        ts_us = t + (int(rng.integers(-jitter, jitter + 1)) if jitter else 0)
        rec[i] = (dev, 0, ts_us, dur, freq, avg)
        name = devices[dev] if dev < n_dev else "ghost"
        ref.add(radiotracking.Signal(name, rtm.us_to_datetime(ts_us), freq, dur * US, avg + 3, avg, 1.0, -100.0, 10.0))
    # native: the same stream in ragged batches
    got_ts, got_dur, got_freq, got_avgs = [], [], [], []
    i = 0
    while i < n:
        k = int(rng.choice([1, 2, 7, 64, n]))
        b = nat.add(rec[i:i + k])
        i += k
        got_ts += [int(x) for x in b.groups["ts_us"]]
        got_dur += [int(x) for x in b.groups["duration_us"]]
        got_freq += [float(x) for x in b.groups["frequency"]]
        got_avgs += [[None if not p else float(a) for a, p in zip(row_a, row_p)] for row_a, row_p in zip(b.avgs, b.present)]
    want = sink.items
    ok = len(want) == len(got_ts)
    if ok:
        for j, r in enumerate(want):
            ok = ok and rtm.datetime_to_us(r.ts) == got_ts[j] and r.duration // US == got_dur[j] and r.frequency == got_freq[j] and list(r._avgs) == got_avgs[j]
    pend = nat.pending()
    ok = ok and len(pend) == len(ref._matched)
    if ok:
        for j, r in enumerate(ref._matched):
            ok = ok and rtm.datetime_to_us(r.ts) == int(pend.groups["ts_us"][j]) and r.frequency == float(pend.groups["frequency"][j])
    n_cases += 1
    n_sig += n
    n_groups += len(want)
    if not ok:
        n_bad += 1
        print(f"MISMATCH case {case}: devices {n_dev} params {params} signals {n}: {len(got_ts)} vs {len(want)} groups, pending {len(pend)} vs {len(ref._matched)}", flush=True)
    nat.close()
print(f"SOAK MATCH: {n_cases} cases, {n_sig} signals, {n_groups} consumed groups, {n_bad} mismatching cases")
