"""Runner integration: device -> (GPU, slot) plan, per-stream life cycle with the
reference's rules (radiotracking/__main__.py:94-190, analyze.py:180-231).  Host logic on CPU with a
recording stand-in for the batch analyzer; the GPU test drives the real kernels against the oracle."""
import datetime

import numpy as np
import pytest

from pyradiotracking_amd import Signal, StateMessage, _native
from pyradiotracking_amd.analyze import _RecordDecoder
from pyradiotracking_amd.runner import BatchRunner, plan_devices

FS, NPERSEG = 2048000, 256
T0 = 1_700_000_000.0


class Q:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)

    def states(self, device=None):
        return [(m.device, m.state.name) for m in self.items if isinstance(m, StateMessage) and device in (None, m.device)]


class FakeBatch:
    """Records the calls; returns the records planted in ``next_records``."""

    created = []

    def __init__(self, devices, calibration_db=0.0, gpu=0, sample_rate=FS, fft_nperseg=NPERSEG, center_freq=150150000, **kw):
        self.devices, self.calibration_db, self.gpu, self.kw = list(devices), list(calibration_db), gpu, kw
        self.decoder = _RecordDecoder(fft_nperseg, sample_rate, center_freq, self.calibration_db)
        self.enqueued, self.resets, self.closed = [], [], False
        self.pending, self.fail_enqueue, self.fail_fetch = 0, False, False
        self.next_records = np.zeros(0, dtype=_native.RECORD_DTYPE)
        FakeBatch.created.append(self)

    def enqueue(self, chunk):
        if self.fail_enqueue:
            raise RuntimeError(f"GPU {self.gpu} refuses")
        self.enqueued.append(np.array(chunk))
        self.pending += 1

    def fetch_records(self, allow_truncated=False):
        assert self.pending > 0, "fetch without a pending call"
        self.pending -= 1
        if self.fail_fetch:
            raise RuntimeError(f"GPU {self.gpu} has no result")
        return self.next_records

    def reset_stream(self, slot):
        self.resets.append(slot)

    def close(self):
        self.closed = True


def _runner(q, devices=("a", "b", "c"), gpus=(0,), **kw):
    FakeBatch.created = []
    r = BatchRunner(device=devices, gpus=gpus, signal_queue=q, analyzer_factory=FakeBatch, sample_rate=FS, fft_nperseg=NPERSEG, **kw)
    r.start_analyzers()
    return r


def test_plan_devices_blocks_and_calibration_check():
    assert plan_devices(list("abcde"), [0.0] * 5, [0, 1]) == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)]
    assert plan_devices(["x"], [1.5], [3]) == [(3, 0)]
    with pytest.raises(ValueError, match="do not match devices"):
        plan_devices(["0", "1"], [0.0], [0])
    with pytest.raises(ValueError):
        plan_devices(["0"], [0.0], [])
    q = Q()
    r = _runner(q, devices=list("abcde"), gpus=(0, 1), calibration=[1, 2, 3, 4, 5])
    assert [(b.gpu, b.devices, b.calibration_db) for b in FakeBatch.created] == [(0, ["a", "b", "c"], [1.0, 2.0, 3.0]), (1, ["d", "e"], [4.0, 5.0])]
    assert BatchRunner(device=["0", "1"], analyzer_factory=FakeBatch).calibration == [0.0, 0.0]  # __main__.py:215-217
    r.stop_analyzers()
    assert all(b.closed for b in FakeBatch.created)


def test_heartbeats_and_signal_routing():
    q = Q()
    r = _runner(q, state_update_s=5)
    blen = FS
    buf = np.zeros((3, blen), np.complex64)
    rec = np.zeros(3, dtype=_native.RECORD_DTYPE)
    rec["stream"] = [0, 2, 2]
    rec["fi"] = [3, 4, 5]
    rec["start"] = [10, 20, -2]
    rec["end"] = [100, 120, 90]
    rec["max_p"] = rec["mean_p"] = rec["row_mean"] = 1e-8
    rec["shadowed"] = [0, 0, 1]
    FakeBatch.created[0].next_records = rec
    assert r.process(buf, now=T0) == 2  # the shadowed record is not published (analyze.py:248-251)
    sigs = [m for m in q.items if isinstance(m, Signal)]
    assert [s.device for s in sigs] == ["a", "c"]
    ts0 = datetime.datetime.fromtimestamp(T0) - datetime.timedelta(seconds=1)  # ts_start = _ts - buffer length (:231)
    want = (ts0 + datetime.timedelta(seconds=(NPERSEG / 2 + 10 * NPERSEG) / FS)).astimezone(datetime.timezone.utc)
    assert sigs[0].ts == want
    assert q.states() == [("a", "STARTED"), ("b", "STARTED"), ("c", "STARTED")]
    q.items.clear()
    FakeBatch.created[0].next_records = rec[:0]
    for k in range(1, 8):
        r.process(buf, now=T0 + k)
    # RUNNING at once (a different state), then again only after state_update_s (analyze.py:180-190)
    assert q.states("a") == [("a", "RUNNING"), ("a", "RUNNING")]
    assert [m.ts for m in q.items if m.device == "a"] == [
        datetime.datetime.fromtimestamp(T0 + 1).astimezone(datetime.timezone.utc),
        datetime.datetime.fromtimestamp(T0 + 7).astimezone(datetime.timezone.utc),
    ]


def test_timeout_restart_budget_and_termination():
    q = Q()
    r = _runner(q, sdr_max_restart=1, sdr_timeout_s=2)
    fake = FakeBatch.created[0]
    buf = np.ones((3, FS), np.complex64)
    r.check_analyzers(now=T0)  # nothing has started: nothing happens (:163)
    assert q.items == []
    r.process(buf, now=T0)
    r.process(buf, present=[True, False, True], now=T0 + 1)
    assert np.all(fake.enqueued[-1][1] == 0) and np.all(fake.enqueued[-1][0] == 1)  # the absent SDR's row is blanked
    r.check_analyzers(now=T0 + 1.5)
    assert r.streams[1].restarts == 0
    r.process(buf, present=[True, False, True], now=T0 + 2)
    q.items.clear()
    r.check_analyzers(now=T0 + 2.5)  # b: last data at T0, older than 2 s -> STOPPED with that time stamp, restart
    assert [(m.device, m.state.name, m.ts.timestamp()) for m in q.items] == [("b", "STOPPED", T0)]
    assert fake.resets == [1] and r.streams[1].sdr_max_restart == 0 and r.streams[1].restarts == 1 and r.running
    assert r.streams[1].last_data_ts == 0.0 and r.streams[1].ts is None
    q.items.clear()
    r.process(buf, now=T0 + 3)  # b is back: a fresh analyzer reports STARTED again
    assert q.states("b") == [("b", "STARTED")]
    # b dies again: the budget is used up -> the whole station terminates (:180-183), every stream reports STOPPED
    r.process(buf, present=[True, False, True], now=T0 + 4)
    r.process(buf, present=[True, False, True], now=T0 + 5)
    r.process(buf, present=[True, False, True], now=T0 + 6)
    q.items.clear()
    r.check_analyzers(now=T0 + 6.5)
    assert not r.running and fake.closed
    assert q.states() == [("b", "STOPPED"), ("a", "STOPPED"), ("b", "STOPPED"), ("c", "STOPPED")]
    with pytest.raises(RuntimeError):
        r.process(buf, now=T0 + 7)


def test_clock_drift_stops_a_stream_and_a_gap_drops_the_look_back():
    q = Q()
    r = _runner(q, sdr_timeout_s=100)
    fake = FakeBatch.created[0]
    buf = np.zeros((3, FS), np.complex64)
    r.process(buf, now=T0)
    r.process(buf, now=T0 + 1)
    q.items.clear()
    # stream c's buffer arrives 2.5 buffer lengths late: total drift > two blocks (analyze.py:226-229)
    r.process(buf, present=[True, True, False], now=T0 + 2)
    r.process(buf, present=[True, True, False], now=T0 + 3)
    assert fake.resets == []
    r.process(buf, now=T0 + 4.6)
    assert fake.resets == [2]  # c missed steps: what the handle holds is not the buffer before this one
    assert ("c", "STOPPED") in q.states("c") and not r.streams[2].alive
    assert r.streams[0].alive and r.streams[1].alive
    r.check_analyzers(now=T0 + 5)
    assert r.streams[2].alive and r.streams[2].restarts == 1 and fake.resets == [2, 2]


@pytest.mark.gpu
def test_runner_on_the_gpu_matches_per_sdr_oracles():
    """Four SDRs with their own calibrations on one GPU; one of them drops out for a step and is later
    restarted after a time-out.  Every published Signal equals what a per-SDR CPU analyzer (the oracle,
    restarted at the same moments) produces."""
    from oracle import analyze_oracle as oracle
    from pyradiotracking_amd import synth

    if _native.device_count() < 1:
        pytest.fail("no GPU visible")
    fs, nperseg = 2048000, 256
    blen, n_buf = 400 * nperseg, 6
    cal = [0.0, 4.0, -3.0, 1.5]
    w = oracle.window_coefficients("hamming", nperseg)
    rng = np.random.default_rng(11)
    iq = []
    for s in range(4):
        pulses = synth.random_pulses(rng, n_buf * blen, fs, w, 30, dur_ms=(9, 30), peak_dbw=(-95.0, -70.0))
        for k in range(1, n_buf):
            pulses.append(synth.Pulse(k * blen - int(0.006 * fs), int(0.015 * fs), (0.1 + 0.07 * s) * fs, synth.amp_for_peak_dbw(-66.0, w, fs), 0.5))
        iq.append(synth.make_stream(synth.StreamSpec(n_buf * blen, fs, pulses), 40 + s))
    iq = np.stack(iq)
    q = Q()
    kw = dict(sample_rate=fs, fft_nperseg=nperseg)
    r = BatchRunner(device=["0", "1", "2", "3"], calibration=cal, gpus=[0], sdr_timeout_s=0.1, signal_queue=q,
                    sdr_callback_length=blen, **kw)
    r.start_analyzers()
    oas = [oracle.OracleAnalyzer(device=str(s), calibration_db=cal[s], **kw) for s in range(4)]
    dt = blen / fs
    present = {1: [True, True, False, True], 3: [True, False, True, True], 4: [True, False, True, True]}
    want = []
    clocks = [None] * 4
    for k in range(n_buf):
        now = T0 + k * dt
        pres = present.get(k, [True] * 4)
        if k == 5:
            r.check_analyzers(now=now - 0.01)  # SDR 1 has been silent for two steps: restarted
            oas[1].reset()
            clocks[1] = None
            assert r.streams[1].restarts == 1
        if k == 2:
            oas[2].reset()  # its previous buffer is not the one before this one
        chunk = np.ascontiguousarray(iq[:, k * blen:(k + 1) * blen])
        for s in range(4):
            if not pres[s]:
                continue
            recv = datetime.datetime.fromtimestamp(now)
            clocks[s] = recv if clocks[s] is None else clocks[s] + datetime.timedelta(seconds=dt)
            _, kept = oas[s].process(chunk[s], clocks[s] - datetime.timedelta(seconds=dt))
            want += kept
        r.process(chunk, present=pres, now=now)
    got = [m for m in q.items if isinstance(m, Signal)]
    assert len(got) == len(want) > 20
    for g, x in zip(got, want):
        assert (g.device, g.ts, g.duration, g.frequency) == (x.device, x.ts, x.duration, x.frequency)
        for name in ("max", "avg", "noise", "snr", "std"):
            assert abs(getattr(g, name) - getattr(x, name)) < 0.01
    assert q.states("1") == [("1", "STARTED"), ("1", "RUNNING"), ("1", "STOPPED"), ("1", "STARTED")]
    r.stop_analyzers()


def test_a_failing_gpu_does_not_leave_the_station_one_step_behind():
    """enqueue on every GPU, then fetch -- a failure on one GPU must not leave a call pending on another
    (rt_fetch is FIFO: every later step would get the records of the step before), and a step that was never
    enqueued must not advance the stream clocks."""
    q = Q()
    r = _runner(q, devices=list("abcd"), gpus=(0, 1))
    g0, g1 = FakeBatch.created
    buf = np.zeros((4, FS), np.complex64)
    # (1) GPU 1 refuses the step: GPU 0's call is drained, no clock has moved, no heartbeat was sent
    g1.fail_enqueue = True
    with pytest.raises(RuntimeError, match="GPU 1 refuses"):
        r.process(buf, now=T0)
    assert g0.pending == 0 and g1.pending == 0
    assert all(st.ts is None and st.last_data_ts == 0.0 for st in r.streams) and q.items == []
    g1.fail_enqueue = False
    assert r.process(buf, now=T0) == 0
    assert q.states() == [(d, "STARTED") for d in "abcd"]
    # (2) GPU 0 has no result for a step: GPU 1 is fetched all the same, its signals are published, then the error surfaces
    rec = np.zeros(1, dtype=_native.RECORD_DTYPE)
    rec["start"], rec["end"] = 5, 80
    rec["max_p"] = rec["mean_p"] = rec["row_mean"] = 1e-8
    g1.next_records = rec
    g0.fail_fetch = True
    q.items.clear()
    with pytest.raises(RuntimeError, match="GPU 0 has no result"):
        r.process(buf, now=T0 + 1)
    assert g0.pending == 0 and g1.pending == 0
    assert [m.device for m in q.items if isinstance(m, Signal)] == ["c"]
    # (3) the next step is in step on both GPUs
    g0.fail_fetch = False
    q.items.clear()
    assert r.process(buf, now=T0 + 2) == 1 and g0.pending == 0 and g1.pending == 0


@pytest.mark.gpu
def test_a_noisy_sdr_degrades_instead_of_aborting_the_step():
    """Two analyzers (two logical GPUs, here both on GPU 0) with a tiny record_capacity: one SDR produces more
    signals than fit.  The step goes through, the other SDRs' signals are complete, the noisy one keeps
    record_capacity of its own, and the following steps are not one call behind."""
    from pyradiotracking_amd import synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

    if _native.device_count() < 1:
        pytest.fail("no GPU visible")
    fs, nperseg, blen = 2048000, 256, 900 * 256
    w = window_coefficients("hamming", nperseg)
    rng = np.random.default_rng(21)

    def stream(n_pulses, seed):
        return synth.make_stream(synth.StreamSpec(blen, fs, synth.random_pulses(rng, blen, fs, w, n_pulses, dur_ms=(9, 12), keep_clear_tail=1024)), seed)

    steps = [np.stack([stream(12 if (s == 0 and k == 0) else 1, 100 + 10 * k + s) for s in range(4)]) for k in range(3)]

    def run(record_capacity):
        q = Q()
        r = BatchRunner(device=list("abcd"), gpus=[0, 1], signal_queue=q, sdr_callback_length=blen, sample_rate=fs, fft_nperseg=nperseg,
                        analyzer_factory=lambda devices, gpu=0, **kw: BatchSignalAnalyzer(devices, gpu=0, record_capacity=record_capacity, **kw))
        r.start_analyzers()
        per_step = []
        for k, chunk in enumerate(steps):
            q.items.clear()
            r.process(chunk, now=T0 + k * blen / fs)
            per_step.append([(m.device, m.ts, m.frequency, m.duration) for m in q.items if isinstance(m, Signal)])
        r.stop_analyzers()
        return per_step

    full, small = run(1024), run(4)
    assert len([x for x in full[0] if x[0] == "a"]) >= 2
    assert [x for x in small[0] if x[0] != "a"] == [x for x in full[0] if x[0] != "a"]
    # the noisy SDR keeps at most record_capacity signals (the shadow filter then sees the truncated list)
    assert 0 < len([x for x in small[0] if x[0] == "a"]) <= 4
    assert small[1:] == full[1:] and all(len(x) > 0 for x in full[1:])
