#!/usr/bin/env python3
"""Throughput of the record serialisers: native batch formatting (rt_format_signals on a row array)
against the standard-library path the reference takes per Signal object (csv.writer / json.dumps with
the csvify / jsonify converters, consume.py:141-151).  Host code only.

    python tools/bench_format.py [n_rows]
"""
import csv
import datetime
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyradiotracking_amd import Signal
from pyradiotracking_amd import consume as rtc
from pyradiotracking_amd.match import us_to_datetime

US = datetime.timedelta(microseconds=1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    rng = np.random.default_rng(0)
    names = [str(i) for i in range(16)]
    rows = np.zeros(n, dtype=rtc.SIGNAL_ROW_DTYPE)
    rows["device"] = rng.integers(0, 16, n)
    rows["ts_us"] = 1_700_000_000_000_000 + np.sort(rng.integers(0, 10**10, n))
    rows["duration_us"] = rng.integers(8000, 40000, n)
    rows["frequency"] = 150e6 + rng.integers(-2000, 2000, n) * 1171.875
    for f in ("max_dbw", "avg_dbw", "std_db", "noise_dbw", "snr_db"):
        rows[f] = rng.uniform(-120, -20, n).astype(np.float32)
    res = {"rows": n}
    rtc.format_signals("csv", rows[:1000], names)  # first call binds the library and touches its pages
    for kind in ("csv", "json", "cbor"):
        t0 = time.perf_counter()
        msgs = rtc.format_signals(kind, rows, names)
        dt = time.perf_counter() - t0
        res[f"native_{kind}_rows_per_s"] = round(n / dt)
        res[f"{kind}_bytes_per_row"] = round(len(msgs.data) / n, 1)
    m = min(n, 50000)
    sigs = [Signal(names[r["device"]], us_to_datetime(r["ts_us"]), r["frequency"], int(r["duration_us"]) * US, r["max_dbw"], r["avg_dbw"],
                   r["std_db"], r["noise_dbw"], r["snr_db"]) for r in rows[:m]]
    t0 = time.perf_counter()
    for s in sigs:
        buf = io.StringIO()
        csv.writer(buf, dialect="excel", delimiter=";").writerow([rtc.csvify(v) for v in s.as_list])
        buf.getvalue()
    res["python_csv_rows_per_s"] = round(m / (time.perf_counter() - t0))
    t0 = time.perf_counter()
    for s in sigs:
        json.dumps(s.as_dict, default=rtc.jsonify)
    res["python_json_rows_per_s"] = round(m / (time.perf_counter() - t0))
    res["python_sample"] = m
    print(json.dumps(res))


if __name__ == "__main__":
    main()
