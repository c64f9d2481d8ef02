#!/usr/bin/env python3
"""Throughput of the cross-SDR matcher: native batch path (rt_match_add on record arrays), the
drop-in class fed Signal by Signal, and the CPU restatement of the reference (oracle/match_oracle.py,
the same Python-level algorithm the reference runs).  Host code only.

    python tests/perf/bench_match.py [n_signals] [n_devices] [n_tags]
"""
import datetime
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from oracle.match_oracle import MatchInput, OracleMatcher
from pyradiotracking_amd import Signal
from pyradiotracking_amd import match as rtm

US = datetime.timedelta(microseconds=1)


def stream(n, n_dev, n_tags, seed=0):
    rng = np.random.default_rng(seed)
    f0 = 150e6 + rng.integers(0, 400, n_tags) * 1000.0
    rec = np.zeros(n, dtype=rtm.SIGNAL_DTYPE)
    tag = rng.integers(0, n_tags, n)
    rec["device"] = rng.integers(0, n_dev, n)
    rec["ts_us"] = 1_700_000_000_000_000 + np.sort(rng.integers(0, n * 2500, n))  # ~400 signals per second
    rec["duration_us"] = 20000 + rng.integers(-500, 500, n)
    rec["frequency"] = f0[tag] + rng.uniform(-300, 300, n)
    rec["avg"] = rng.uniform(-85, -40, n)
    return rec


class Sink:
    def __init__(self):
        self.n = 0

    def put(self, x):
        self.n += 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    n_dev = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n_tags = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    rec = stream(n, n_dev, n_tags)
    params = dict(matching_timeout_s=2.0, matching_time_diff_s=0.05, matching_bandwidth_hz=4000.0,
                  matching_duration_diff_ms=5.0)
    devices = [str(i) for i in range(n_dev)]

    nm = rtm.NativeMatcher(n_dev, 2.0, 0.05, 4000.0, 5.0)
    t0 = time.perf_counter()
    out = nm.add(rec)
    t_native = time.perf_counter() - t0

    n_py = min(n, 40000)
    sigs = [Signal(str(int(r["device"])), rtm.us_to_datetime(r["ts_us"]), float(r["frequency"]), int(r["duration_us"]) * US,
                   0.0, float(r["avg"]), 0.0, 0.0, 0.0) for r in rec[:n_py]]
    sink = Sink()
    m = rtm.SignalMatcher(device=devices, signal_queue=sink, **params)
    t0 = time.perf_counter()
    for s in sigs:
        m.add(s)
    t_class = time.perf_counter() - t0

    om = OracleMatcher(devices, **params)
    ins = [MatchInput(s.device, s.ts, s.frequency, s.duration, s.avg) for s in sigs]
    t0 = time.perf_counter()
    k = 0
    for x in ins:
        k += len(om.add(x))
    t_oracle = time.perf_counter() - t0

    print(json.dumps({
        "signals": n, "devices": n_dev, "tags": n_tags, "groups_consumed": len(out),
        "native_batch_signals_per_s": round(n / t_native), "drop_in_class_signals_per_s": round(n_py / t_class),
        "cpu_restatement_signals_per_s": round(n_py / t_oracle), "python_sample": n_py,
        "speedup_batch_vs_cpu_restatement": round((n / t_native) / (n_py / t_oracle), 1),
    }))


if __name__ == "__main__":
    main()
