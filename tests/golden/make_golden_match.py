#!/usr/bin/env python3
"""Generate tests/golden/match_cases.npz by running the REFERENCE matcher.

Runs only in the build container (needs /root/reference).  ``radiotracking.match``
imports ``radiotracking.consume``, which imports ``cbor2`` and ``paho.mqtt.client``
(neither installed); empty stand-in modules are injected for the import only --
the matcher never calls into them (match.py:54-82 uses datetime arithmetic and a
queue's ``put``).

Per case the file holds the inputs (matcher parameters, the signal stream as
arrays: device index, ts in microseconds since the epoch, frequency, duration in
microseconds, avg) and what the reference did with them: how many groups each
``add`` consumed, every consumed group in queue order (ts, frequency, duration,
per-device avgs with a presence mask) and the groups still open at the end.

Usage:  TZ=UTC python tests/golden/make_golden_match.py
"""
import datetime
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

for name in ("cbor2", "paho", "paho.mqtt", "paho.mqtt.client"):
    sys.modules.setdefault(name, types.ModuleType(name))

import radiotracking  # noqa: E402
import radiotracking.match as ref_match  # noqa: E402

EPOCH = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)
US = datetime.timedelta(microseconds=1)
T0_US = 1704067200 * 10**6  # 2024-01-01T00:00:00Z


class Sink:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)


def to_us(ts):
    return (ts - EPOCH) // US


def make_stream(rng, n_dev, n_tags, seconds, detect_p, unknown_dev, dup_p, jitter_us, freq_jitter, dur_jitter_us,
                disorder_us):
    """Pulsed tags seen by several SDRs: (arrival key, device, ts_us, freq, dur_us, avg) rows in arrival order."""
    rows = []
    for _ in range(n_tags):
        f0 = 150.0e6 + float(rng.integers(0, 200)) * 1000.0 + float(rng.uniform(-300, 300))
        dur = int(rng.integers(8000, 40000))
        period = float(rng.uniform(0.4, 1.6))
        t = float(rng.uniform(0, period))
        while t < seconds:
            base_ts = T0_US + int(t * 1e6)
            for d in range(n_dev + (1 if unknown_dev else 0)):
                if rng.uniform() > detect_p:
                    continue
                reps = 2 if rng.uniform() < dup_p else 1
                for _ in range(reps):
                    ts = base_ts + int(rng.integers(-jitter_us, jitter_us + 1))
                    fr = f0 + float(rng.choice([-1.0, 0.0, 0.0, 1.0])) * freq_jitter + float(rng.uniform(-1, 1))
                    du = dur + int(rng.integers(-dur_jitter_us, dur_jitter_us + 1))
                    avg = float(rng.uniform(-85, -40))
                    key = ts + int(rng.integers(-disorder_us, disorder_us + 1))
                    rows.append((key, d, ts, fr, du, avg))
            t += period
    rows.sort(key=lambda r: r[0])
    return rows


def run_case(name, rng, n_dev, params, **stream_kw):
    devices = [str(i) for i in range(n_dev)]
    sink = Sink()
    matcher = ref_match.SignalMatcher(device=devices, signal_queue=sink, **params)
    rows = make_stream(rng, n_dev, **stream_kw)
    n = len(rows)
    dev = np.array([r[1] for r in rows], dtype=np.int32)
    ts_us = np.array([r[2] for r in rows], dtype=np.int64)
    freq = np.array([r[3] for r in rows], dtype=np.float64)
    dur_us = np.array([r[4] for r in rows], dtype=np.int64)
    avg = np.array([r[5] for r in rows], dtype=np.float64)
    emitted_per_add = np.zeros(n, dtype=np.int32)
    for i in range(n):
        name_i = devices[dev[i]] if dev[i] < n_dev else "ghost"
        sig = radiotracking.Signal(name_i, EPOCH + int(ts_us[i]) * US, float(freq[i]),
                                   datetime.timedelta(microseconds=int(dur_us[i])), float(avg[i]) + 3.0, float(avg[i]),
                                   1.0, -100.0, 10.0)
        before = len(sink.items)
        matcher.add(sig)
        emitted_per_add[i] = len(sink.items) - before

    def pack(groups):
        g_ts = np.array([to_us(g.ts) for g in groups], dtype=np.int64)
        g_freq = np.array([g.frequency for g in groups], dtype=np.float64)
        g_dur = np.array([g.duration // US for g in groups], dtype=np.int64)
        g_avgs = np.full((len(groups), n_dev), np.nan, dtype=np.float64)
        g_present = np.zeros((len(groups), n_dev), dtype=np.uint8)
        g_members = np.array([len(g._sigs) for g in groups], dtype=np.int32)
        for k, g in enumerate(groups):
            for d, a in enumerate(g._avgs):
                if a is not None:
                    g_avgs[k, d] = a
                    g_present[k, d] = 1
        return g_ts, g_freq, g_dur, g_avgs, g_present, g_members

    out = {}
    pre = f"{name}/"
    out[pre + "n_dev"] = np.int32(n_dev)
    out[pre + "timeout_s"] = np.float64(params["matching_timeout_s"])
    out[pre + "time_diff_s"] = np.float64(params["matching_time_diff_s"])
    out[pre + "bandwidth_hz"] = np.float64(params["matching_bandwidth_hz"])
    dd = params.get("matching_duration_diff_ms")
    out[pre + "duration_diff_ms"] = np.float64(np.nan if dd is None else dd)
    for key, arr in (("dev", dev), ("ts_us", ts_us), ("freq", freq), ("dur_us", dur_us), ("avg", avg),
                     ("emitted_per_add", emitted_per_add)):
        out[pre + key] = arr
    for tag, groups in (("out", sink.items), ("open", list(matcher._matched))):
        for key, arr in zip(("ts_us", "freq", "dur_us", "avgs", "present", "members"), pack(groups)):
            out[pre + f"{tag}_{key}"] = arr
    # a few repr / str strings of consumed groups (format parity of the result type)
    out[pre + "out_repr"] = np.array([repr(g) for g in sink.items[:6]])
    out[pre + "out_str"] = np.array([str(g) for g in sink.items[:6]])
    print(f"{name}: {n} signals -> {len(sink.items)} consumed, {len(matcher._matched)} open")
    return out


def main():
    rng = np.random.default_rng(20240101)
    base = dict(n_tags=6, seconds=20.0, detect_p=0.8, unknown_dev=False, dup_p=0.0, jitter_us=300, freq_jitter=1171.875,
                dur_jitter_us=900, disorder_us=0)
    cases = {}
    cases.update(run_case("defaults", rng, 4, dict(matching_timeout_s=2.0, matching_time_diff_s=0.0,
                                                   matching_bandwidth_hz=0.0), **base))
    cases.update(run_case("tolerant", rng, 4, dict(matching_timeout_s=2.0, matching_time_diff_s=0.05,
                                                   matching_bandwidth_hz=4000.0), **base))
    cases.update(run_case("duration", rng, 4, dict(matching_timeout_s=1.0, matching_time_diff_s=0.01,
                                                   matching_bandwidth_hz=2500.0, matching_duration_diff_ms=3.001),
                          **base))
    cases.update(run_case("duration_zero_us", rng, 3, dict(matching_timeout_s=1.0, matching_time_diff_s=0.01,
                                                           matching_bandwidth_hz=2500.0,
                                                           matching_duration_diff_ms=0.0005), **base))
    cases.update(run_case("odd_halves", rng, 4, dict(matching_timeout_s=0.7500005, matching_time_diff_s=0.0000015,
                                                     matching_bandwidth_hz=2343.75, matching_duration_diff_ms=1.001),
                          **{**base, "jitter_us": 3, "dur_jitter_us": 2}))
    cases.update(run_case("replace_and_ghost", rng, 3, dict(matching_timeout_s=2.0, matching_time_diff_s=0.02,
                                                            matching_bandwidth_hz=3000.0),
                          **{**base, "unknown_dev": True, "dup_p": 0.4}))
    cases.update(run_case("disordered", rng, 4, dict(matching_timeout_s=0.5, matching_time_diff_s=0.02,
                                                     matching_bandwidth_hz=3000.0, matching_duration_diff_ms=4.0),
                          **{**base, "disorder_us": 400000, "n_tags": 10}))
    cases.update(run_case("crowded", rng, 6, dict(matching_timeout_s=5.0, matching_time_diff_s=0.1,
                                                  matching_bandwidth_hz=6000.0),
                          **{**base, "n_tags": 40, "seconds": 10.0}))
    np.savez_compressed(os.path.join(HERE, "match_cases.npz"), **cases)


if __name__ == "__main__":
    main()
