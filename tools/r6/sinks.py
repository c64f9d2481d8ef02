#!/usr/bin/env python3
"""Host sinks on the GPU box's cores (no GPU needed): rows from records, CSV / JSON formatting and the matcher fleet by thread count.
usage: python tools/r6/sinks.py"""
import ctypes as C, datetime, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyradiotracking_amd import _native, consume, match as rtm
from pyradiotracking_amd.analyze import _RecordDecoder
from pyradiotracking_amd.match import datetime_to_us

n, S = 600000, 4096
rng = np.random.default_rng(0)
rec = np.zeros(n, dtype=_native.RECORD_DTYPE)
rec["stream"] = np.sort(rng.integers(0, S, n)); rec["fi"] = rng.integers(0, 256, n)
rec["start"] = rng.integers(0, 1100, n); rec["end"] = rec["start"] + rng.integers(9, 40, n)
rec["max_p"] = rng.uniform(1e-9, 1e-6, n); rec["mean_p"] = rec["max_p"] * 0.7; rec["row_mean"] = 1e-12; rec["std_db"] = rng.uniform(5, 20, n)
dec = _RecordDecoder(256, 300000, 150150000, 0.0)
names = [str(i) for i in range(S)]
ts0 = [datetime_to_us(datetime.datetime(2024, 1, 1))] * S

def best(f, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t0)
    return b, r

print("cores", len(os.sched_getaffinity(0)))
# the matcher fleet: stations of four SDRs, a batch of signals per station in time order
n_st, nd = S // 4, 4
sig_rows = consume.rows_from_analysis(rec, dec, ts0)
st = sig_rows["device"] // nd
order = np.lexsort((sig_rows["ts_us"], st))
msig = np.zeros(len(order), dtype=rtm.SIGNAL_DTYPE)
msig["device"] = sig_rows["device"][order] % nd; msig["ts_us"] = sig_rows["ts_us"][order]; msig["duration_us"] = sig_rows["duration_us"][order]
msig["frequency"] = sig_rows["frequency"][order]; msig["avg"] = sig_rows["avg_dbw"][order]
offs = np.searchsorted(st[order], np.arange(n_st + 1))
for th in (1, 2, 4, 8, 16, 32):
    consume.set_host_threads(th)
    dt, rows = best(lambda: consume.rows_from_analysis(rec, dec, ts0))
    out = [f"threads {th:2d}: rows_from_analysis {n / dt / 1e6:6.2f} M/s"]
    for kind in ("csv", "json", "cbor"):
        dt, _ = best(lambda: consume.format_signals(kind, rows, names))
        out.append(f"{kind} {n / dt / 1e6:6.2f} M/s")
    dt, _ = best(lambda: consume.format_signals("csv", consume.rows_from_analysis(rec, dec, ts0), names))
    out.append(f"records->csv {n / dt / 1e6:6.2f} M/s")
    def fleet_run():
        fl = rtm.MatcherFleet(n_st, nd, timeout_s=2.0, time_diff_s=0.0, bandwidth_hz=4000.0)
        t0 = time.perf_counter(); fl.add(msig, offs); dt = time.perf_counter() - t0
        fl.close()
        return dt
    dtm = min(fleet_run() for _ in range(3))
    out.append(f"matcher fleet ({n_st} stations x {nd}) {len(msig) / dtm / 1e6:6.2f} M signals/s")
    print("  ".join(out), flush=True)
consume.set_host_threads(0)
