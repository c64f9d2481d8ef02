#!/usr/bin/env python3
"""Randomised soak of BatchRunner (device plan, per-SDR calibration, drop-outs, time-outs, restart budget, clock
drift) on the GPU against per-SDR oracles driven by the same rules: every Signal and StateMessage on the queue
is compared.  usage: soak_runner.py [seconds] [seed]"""
import datetime
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import analyze_oracle as oracle  # noqa: E402
from pyradiotracking_amd import Signal, StateMessage, synth  # noqa: E402
from pyradiotracking_amd.runner import BatchRunner  # noqa: E402

T0 = 1_700_000_000.0


class Q:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
n_cases = n_sig = n_bad = 0
case = 0
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed0, case])
    fs, nperseg = int(rng.choice([300000, 2048000])), int(rng.choice([256, 1024]))
    n_dev = int(rng.integers(1, 7))
    blen = int(rng.integers(60, 300)) * nperseg
    n_steps = int(rng.integers(3, 9))
    cal = [float(c) for c in rng.uniform(-5, 5, n_dev)]
    dt = blen / fs
    timeout_s = float(rng.choice([0.6, 1.5, 2.5])) * dt
    max_restart = int(rng.integers(0, 3))
    w = oracle.window_coefficients("hamming", nperseg)
    hop = nperseg / fs
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, signal_min_duration_ms=8 * hop * 1e3, signal_max_duration_ms=60 * hop * 1e3)
    iq = []
    for s in range(n_dev):
        total = n_steps * blen
        pulses = synth.random_pulses(rng, total, fs, w, int(rng.integers(4, 20)), dur_ms=(10 * hop * 1e3, 50 * hop * 1e3), peak_dbw=(-92.0, -65.0))
        for k in range(1, n_steps):
            pulses.append(synth.Pulse(k * blen - 10 * nperseg, 25 * nperseg, float(rng.uniform(-0.4, 0.4) * fs), synth.amp_for_peak_dbw(-66.0, w, fs), 0.2))
        iq.append(synth.make_stream(synth.StreamSpec(total, fs, pulses), 100 * case + s))
    iq = np.stack(iq)
    q = Q()
    r = BatchRunner(device=[str(i) for i in range(n_dev)], calibration=cal, gpus=[0], sdr_max_restart=max_restart, sdr_timeout_s=timeout_s,
                    state_update_s=float(rng.choice([0.0, 2.5 * dt, 300.0])), signal_queue=q, sdr_callback_length=blen, **kw)
    r.start_analyzers()
    # the model: per-SDR oracle + the reference's rules (analyze.py:180-231, __main__.py:153-190)
    oas = [oracle.OracleAnalyzer(device=str(s), calibration_db=cal[s], **kw) for s in range(n_dev)]
    st = [dict(last=0.0, ts=None, state=None, alive=True, stale=False, budget=max_restart) for _ in range(n_dev)]
    want = []
    su = r.state_update_s

    def model_state(s, ts, state):
        ts = ts.astimezone(datetime.timezone.utc)
        last = st[s]["state"]
        if last and last[1] == state and last[0] + datetime.timedelta(seconds=su) >= ts:
            return
        st[s]["state"] = (ts, state)
        want.append(("state", str(s), ts, state))

    now = T0
    terminated = False
    for k in range(n_steps):
        now += dt * float(rng.choice([1.0, 1.0, 1.0, 1.02, 3.2]))  # sometimes a late step (clock drift beyond two blocks)
        present = [bool(rng.random() < 0.8) for _ in range(n_dev)]
        chunk = np.ascontiguousarray(iq[:, k * blen:(k + 1) * blen])
        # model first
        for s in range(n_dev):
            m = st[s]
            if not (present[s] and m["alive"]):
                m["stale"] = True
                continue
            if m["stale"]:
                oas[s].reset()
                m["stale"] = False
            recv = datetime.datetime.fromtimestamp(now)
            blen_dt = datetime.timedelta(seconds=blen / fs)
            model_state(s, recv, StateMessage.State.STARTED if not m["last"] else StateMessage.State.RUNNING)
            m["last"] = datetime.datetime.timestamp(recv)
            m["ts"] = recv if not m["ts"] else m["ts"] + blen_dt
            if (recv - m["ts"]).total_seconds() > 2 * blen_dt.total_seconds():
                model_state(s, recv, StateMessage.State.STOPPED)
                m["alive"] = False
            m["pending"] = m["ts"] - blen_dt
        for s in range(n_dev):
            m = st[s]
            if "pending" in m:
                _, kept = oas[s].process(chunk[s], m.pop("pending"))
                want += [("signal", x) for x in kept]
        r.process(chunk, present=present, now=now)
        if rng.random() < 0.7:
            chk = now + float(rng.uniform(0, 0.9)) * dt
            for s in range(n_dev):
                m = st[s]
                if m["alive"]:
                    if m["last"] == 0.0 or m["last"] > chk - timeout_s:
                        continue
                    want.append(("state", str(s), datetime.datetime.fromtimestamp(m["last"], tz=datetime.timezone.utc), StateMessage.State.STOPPED))
                if m["budget"] <= 0:
                    terminated = True
                    for s2 in range(n_dev):
                        want.append(("state", str(s2), datetime.datetime.fromtimestamp(st[s2]["last"], tz=datetime.timezone.utc), StateMessage.State.STOPPED))
                    break
                m.update(budget=m["budget"] - 1, last=0.0, ts=None, state=None, alive=True, stale=False)
                oas[s].reset()
            r.check_analyzers(now=chk)
            if terminated:
                break
    assert terminated == (not r.running)
    got = []
    for m in q.items:
        if isinstance(m, StateMessage):
            got.append(("state", m.device, m.ts, m.state))
        elif isinstance(m, Signal):
            got.append(("signal", m))
    ok = len(got) == len(want)
    if ok:
        for g, x in zip(got, want):
            if g[0] != x[0]:
                ok = False
            elif g[0] == "state":
                ok = ok and g[1:] == x[1:]
            else:
                a, b = g[1], x[1]
                ok = ok and (a.device, a.ts, a.duration, a.frequency) == (b.device, b.ts, b.duration, b.frequency)
                ok = ok and all(abs(getattr(a, f) - getattr(b, f)) < 0.02 for f in ("max", "avg", "std", "noise", "snr"))
            if not ok:
                print(f"  first difference: got {g} want {x}")
                break
    n_cases += 1
    n_sig += sum(1 for x in want if x[0] == "signal")
    if not ok:
        n_bad += 1
        print(f"MISMATCH case {case}: devices {n_dev} steps {n_steps} restart budget {max_restart} timeout {timeout_s:.3f}s: {len(got)} vs {len(want)} messages", flush=True)
    if r.running:
        r.stop_analyzers()
print(f"SOAK RUNNER: {n_cases} cases, {n_sig} signals, {n_bad} mismatching cases")
