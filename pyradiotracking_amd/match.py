"""Cross-SDR signal matcher on record arrays.

Host-side mirror of the reference's ``SignalMatcher`` (radiotracking/match.py:12-82): same
constructor keywords, ``add(signal)`` puts timed-out groups on ``signal_queue`` as
``MatchingSignal`` objects.  The grouping itself runs in the native library
(include/rt_match.h, csrc/rt_match.cpp); ``add_records`` is the batch entry that
takes the analysis path's output as arrays and never builds Python objects per signal.

There is no Python fallback: without ``librt_analyze.so`` the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import datetime as _dt
import math
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import MatchingSignal, Signal
from . import _native

EPOCH = _dt.datetime(1970, 1, 1, tzinfo=_dt.timezone.utc)
_US = _dt.timedelta(microseconds=1)

MATCH_SYMBOLS = (
    "rt_match_create",
    "rt_match_destroy",
    "rt_match_reset",
    "rt_match_pending_count",
    "rt_match_pending_count_many",
    "rt_match_add_many",
    "rt_match_add",
    "rt_match_pending",
    "rt_match_has_member",
    "rt_match_last_error",
)


class RtMatchConfig(C.Structure):
    _fields_ = [
        ("n_devices", C.c_int32),
        ("reserved", C.c_int32),
        ("timeout_s", C.c_double),
        ("time_diff_s", C.c_double),
        ("bandwidth_hz", C.c_double),
        ("duration_diff_ms", C.c_double),
    ]


# include/rt_match.h: rt_match_signal (40 B) and rt_matched (32 B)
SIGNAL_DTYPE = np.dtype(
    [("device", "<i4"), ("reserved", "<i4"), ("ts_us", "<i8"), ("duration_us", "<i8"), ("frequency", "<f8"), ("avg", "<f8")]
)
MATCHED_DTYPE = np.dtype(
    [("ts_us", "<i8"), ("duration_us", "<i8"), ("frequency", "<f8"), ("n_members", "<i4"), ("reserved", "<i4")]
)

_bound = None


def _lib():
    global _bound
    if _bound is not None:
        return _bound
    lib = _native.load_library()
    vp, sz = C.c_void_p, C.c_size_t
    lib.rt_match_create.argtypes = [C.POINTER(RtMatchConfig), C.POINTER(vp)]
    lib.rt_match_destroy.argtypes = [vp]
    lib.rt_match_destroy.restype = None
    lib.rt_match_reset.argtypes = [vp]
    lib.rt_match_pending_count.argtypes = [vp, C.POINTER(sz)]
    lib.rt_match_add.argtypes = [vp, vp, sz, vp, vp, vp, sz, C.POINTER(sz)]
    lib.rt_match_pending.argtypes = [vp, vp, vp, vp, sz, C.POINTER(sz)]
    lib.rt_match_add_many.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, C.c_int32, vp]
    lib.rt_match_pending_count_many.argtypes = [vp, sz, vp]
    lib.rt_match_has_member.argtypes = [vp, sz, vp]
    lib.rt_match_last_error.argtypes = [vp]
    lib.rt_match_last_error.restype = C.c_char_p
    for name in MATCH_SYMBOLS:
        getattr(lib, name)
    _bound = lib
    return lib


def datetime_to_us(ts: _dt.datetime) -> int:
    """Aware datetime -> whole microseconds since the Unix epoch (exact)."""
    if ts.tzinfo is None:
        ts = ts.replace(tzinfo=_dt.timezone.utc)
    return (ts - EPOCH) // _US


def us_to_datetime(us: int) -> _dt.datetime:
    return EPOCH + int(us) * _US


class MatchedBatch:
    """Groups that came out of one native call: ``groups`` (MATCHED_DTYPE), ``avgs`` [n, n_devices]
    float64 (NaN where absent) and ``present`` [n, n_devices] uint8."""

    __slots__ = ("groups", "avgs", "present")

    def __init__(self, groups: np.ndarray, avgs: np.ndarray, present: np.ndarray):
        self.groups, self.avgs, self.present = groups, avgs, present

    def __len__(self) -> int:
        return len(self.groups)

    def to_signals(self, devices: List[str]) -> List[MatchingSignal]:
        out = []
        for g, a, p in zip(self.groups, self.avgs, self.present):
            avgs = [float(x) if q else None for x, q in zip(a, p)]
            out.append(MatchingSignal.from_aggregate(devices, us_to_datetime(g["ts_us"]), float(g["frequency"]),
                                                     int(g["duration_us"]) * _US, avgs))
        return out


class NativeMatcher:
    """Thin owner of an ``rt_matcher`` handle."""

    def __init__(self, n_devices: int, timeout_s: float, time_diff_s: float, bandwidth_hz: float,
                 duration_diff_ms: Optional[float] = None):
        self._lib = _lib()
        self.n_devices = int(n_devices)
        cfg = RtMatchConfig(
            n_devices=self.n_devices,
            timeout_s=float(timeout_s),
            time_diff_s=float(time_diff_s),
            bandwidth_hz=float(bandwidth_hz),
            duration_diff_ms=float(duration_diff_ms) if duration_diff_ms else math.nan,  # match.py:45 `if x else None`
        )
        self._h = C.c_void_p()
        rc = self._lib.rt_match_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise _native.NativeError(rc, "rt_match_create failed")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_match_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc: int):
        if rc != 0:
            raise _native.NativeError(rc, (self._lib.rt_match_last_error(self._h) or b"").decode())

    def pending_count(self) -> int:
        n = C.c_size_t(0)
        self._check(self._lib.rt_match_pending_count(self._h, C.byref(n)))
        return n.value

    def _alloc(self, cap: int):
        nd = max(1, self.n_devices)
        return (np.zeros(cap, dtype=MATCHED_DTYPE), np.full((cap, nd), np.nan, dtype=np.float64),
                np.zeros((cap, nd), dtype=np.uint8))

    def add(self, sigs: np.ndarray) -> MatchedBatch:
        """``sigs``: SIGNAL_DTYPE array in arrival order.  Returns the groups consumed on the way."""
        sigs = np.ascontiguousarray(sigs, dtype=SIGNAL_DTYPE)
        cap = self.pending_count() + len(sigs)
        groups, avgs, present = self._alloc(max(1, cap))
        n = C.c_size_t(0)
        self._check(self._lib.rt_match_add(self._h, sigs.ctypes.data, len(sigs), groups.ctypes.data, avgs.ctypes.data,
                                           present.ctypes.data, cap, C.byref(n)))
        k = n.value
        return MatchedBatch(groups[:k], avgs[:k, : self.n_devices], present[:k, : self.n_devices])

    def pending(self) -> MatchedBatch:
        cap = self.pending_count()
        groups, avgs, present = self._alloc(max(1, cap))
        n = C.c_size_t(0)
        self._check(self._lib.rt_match_pending(self._h, groups.ctypes.data, avgs.ctypes.data, present.ctypes.data, cap,
                                               C.byref(n)))
        k = n.value
        return MatchedBatch(groups[:k], avgs[:k, : self.n_devices], present[:k, : self.n_devices])

    def has_member(self, index: int, sig: np.ndarray) -> bool:
        sig = np.ascontiguousarray(sig, dtype=SIGNAL_DTYPE).reshape(1)
        rc = self._lib.rt_match_has_member(self._h, index, sig.ctypes.data)
        if rc < 0:
            self._check(rc)
        return bool(rc)

    def reset(self):
        self._check(self._lib.rt_match_reset(self._h))


class MatcherFleet:
    """Many stations' matchers driven together: one ``rt_matcher`` per station (the reference runs one ``SignalMatcher`` per
    station process, match.py:21-50), one native call per batch of records (``rt_match_add_many``) with the stations dealt to the
    library's host threads (``consume.set_host_threads``).  The matching rule is sequential inside a station and independent
    between stations, so the result is exactly what one ``NativeMatcher.add`` per station gives."""

    def __init__(self, n_stations: int, devices_per_station: int, timeout_s: float, time_diff_s: float, bandwidth_hz: float,
                 duration_diff_ms: Optional[float] = None):
        self._lib = _lib()
        self.n_devices = int(devices_per_station)
        self.matchers = [NativeMatcher(self.n_devices, timeout_s, time_diff_s, bandwidth_hz, duration_diff_ms) for _ in range(int(n_stations))]
        self._handles = (C.c_void_p * max(1, len(self.matchers)))(*[m._h for m in self.matchers])

    def __len__(self) -> int:
        return len(self.matchers)

    def close(self):
        for m in self.matchers:
            m.close()
        self.matchers = []

    def add(self, sigs: np.ndarray, station_offsets: np.ndarray):
        """``sigs``: SIGNAL_DTYPE rows grouped by station -- station k's rows are ``sigs[station_offsets[k]:station_offsets[k + 1]]``,
        in that station's arrival order, ``device`` = the column inside the station.  Returns ``(batch, group_offsets)``: all
        stations' consumed groups back to back (``MatchedBatch``) and, per station, where its groups start."""
        n_st = len(self.matchers)
        sigs = np.ascontiguousarray(sigs, dtype=SIGNAL_DTYPE)
        so = np.ascontiguousarray(station_offsets, dtype=np.uintp)
        if so.shape != (n_st + 1,) or int(so[-1]) != len(sigs):
            raise ValueError("station_offsets must hold one offset per station plus the end")
        pend = np.zeros(max(1, n_st), dtype=np.uintp)
        rc = self._lib.rt_match_pending_count_many(self._handles, n_st, pend.ctypes.data)
        if rc != 0:
            raise _native.NativeError(rc, "rt_match_pending_count_many failed")
        caps = pend[:n_st].astype(np.int64) + np.diff(so.astype(np.int64))
        oo = np.zeros(n_st + 1, dtype=np.uintp)
        oo[1:] = np.cumsum(caps)
        cap = max(1, int(oo[-1]))
        nd = max(1, self.n_devices)
        groups = np.zeros(cap, dtype=MATCHED_DTYPE)
        avgs = np.full((cap, nd), np.nan, dtype=np.float64)
        present = np.zeros((cap, nd), dtype=np.uint8)
        n_out = np.zeros(max(1, n_st), dtype=np.uintp)
        rc = self._lib.rt_match_add_many(self._handles, n_st, sigs.ctypes.data, so.ctypes.data, groups.ctypes.data, avgs.ctypes.data,
                                         present.ctypes.data, oo.ctypes.data, self.n_devices, n_out.ctypes.data)
        if rc != 0:
            raise _native.NativeError(rc, "rt_match_add_many failed")
        # compact: every station's groups sit at the start of its own share of the output
        counts = n_out[:n_st].astype(np.int64)
        go = np.zeros(n_st + 1, dtype=np.int64)
        go[1:] = np.cumsum(counts)
        # (row j of the compacted output = row oo[station of j] + (j - go[station of j]) of the padded one)
        st_of = np.repeat(np.arange(n_st, dtype=np.int64), counts)
        keep = oo[:n_st].astype(np.int64)[st_of] + (np.arange(int(go[-1]), dtype=np.int64) - go[:n_st][st_of])
        return MatchedBatch(groups[keep], avgs[keep][:, : self.n_devices], present[keep][:, : self.n_devices]), go


class SignalMatcher:
    """Drop-in for ``radiotracking.match.SignalMatcher`` (match.py:12-82).

    ``add(signal)`` ignores anything that is not a ``Signal`` (match.py:63-64), feeds the native
    matcher and puts every group that timed out on ``signal_queue``.  ``add_records`` does the same
    for whole arrays and returns a :class:`MatchedBatch` (nothing is put on the queue unless
    ``emit=True``)."""

    def __init__(
        self,
        device: List[str],
        matching_timeout_s: float,
        matching_time_diff_s: float,
        matching_bandwidth_hz: float,
        signal_queue,
        matching_duration_diff_ms: Optional[float] = None,
        **kwargs,
    ):
        self.devices = device
        self.matching_timeout = _dt.timedelta(seconds=matching_timeout_s)
        self.matching_time_diff = _dt.timedelta(seconds=matching_time_diff_s)
        self.matching_bandwidth_hz = float(matching_bandwidth_hz)
        self.matching_duration_diff = (_dt.timedelta(milliseconds=matching_duration_diff_ms)
                                       if matching_duration_diff_ms else None)
        self.signal_queue = signal_queue
        self._index: Dict[Any, int] = {}
        for i, name in enumerate(device):
            self._index.setdefault(name, i)
        self._extra = len(device)  # ids for device names outside the list (no power column)
        self.native = NativeMatcher(len(device), matching_timeout_s, matching_time_diff_s, matching_bandwidth_hz,
                                    matching_duration_diff_ms)

    # -- record conversion -------------------------------------------------
    def device_id(self, name) -> int:
        idx = self._index.get(name)
        if idx is None:
            idx = self._index[name] = self._extra
            self._extra += 1
        return idx

    def to_records(self, signals: Iterable[Signal]) -> np.ndarray:
        signals = list(signals)
        rec = np.zeros(len(signals), dtype=SIGNAL_DTYPE)
        for i, s in enumerate(signals):
            rec[i] = (self.device_id(s.device), 0, datetime_to_us(s.ts), s.duration // _US, s.frequency, s.avg)
        return rec

    # -- reference interface -----------------------------------------------
    def add(self, signal) -> None:
        if not isinstance(signal, Signal):
            return
        self._emit(self.native.add(self.to_records([signal])))

    def consume(self, msig: MatchingSignal) -> None:
        """Hand a group on (match.py:50-52).  Open groups live in the native matcher; they leave it
        through ``add`` (time-out) or :meth:`flush`, so this only forwards the object."""
        self.signal_queue.put(msig)

    @property
    def _matched(self) -> List[MatchingSignal]:
        return self.native.pending().to_signals(self.devices)

    # -- batch interface -----------------------------------------------------
    def add_records(self, records: np.ndarray, emit: bool = False) -> MatchedBatch:
        batch = self.native.add(records)
        if emit:
            self._emit(batch)
        return batch

    def add_signals(self, signals: Sequence[Signal], emit: bool = True) -> MatchedBatch:
        return self.add_records(self.to_records(signals), emit=emit)

    def flush(self, emit: bool = True) -> MatchedBatch:
        """Consume every open group in list order (shutdown; the reference leaves them behind)."""
        batch = self.native.pending()
        self.native.reset()
        if emit:
            self._emit(batch)
        return batch

    def _emit(self, batch: MatchedBatch) -> None:
        for msig in batch.to_signals(self.devices):
            self.signal_queue.put(msig)


def records_from_analysis(rec: np.ndarray, decoder, ts_start_us: Sequence[int],
                          device_of_stream: Sequence[int]) -> np.ndarray:
    """rt_record array of one analysis call (all streams) -> SIGNAL_DTYPE rows, without a Python
    object per signal.

    ``decoder`` is the analyzer's ``_RecordDecoder`` (``BatchSignalAnalyzer.decoder``): the float64 /
    float32 expressions are the very ones the per-``Signal`` conversion uses.  ``timedelta(seconds=x)``
    rounding is applied through CPython once per distinct value (start and duration are multiples
    of the hop, so there are few).  Shadowed records are dropped: the reference never enqueues them
    (analyze.py:248-251)."""
    r = rec[rec["shadowed"] == 0]
    out = np.zeros(len(r), dtype=SIGNAL_DTYPE)
    if not len(r):
        return out
    t_start, duration_s, frequency, _max, avg_dbw, _std, _noise, _snr = decoder.decode(r)

    def to_us(x):
        uniq, inv = np.unique(x, return_inverse=True)
        conv = np.array([_dt.timedelta(seconds=float(v)) // _US for v in uniq], dtype=np.int64)
        return conv[inv]

    streams = r["stream"]
    out["device"] = np.asarray(device_of_stream, dtype=np.int32)[streams]
    out["ts_us"] = np.asarray(ts_start_us, dtype=np.int64)[streams] + to_us(t_start)
    out["duration_us"] = to_us(duration_s)
    out["frequency"] = frequency
    out["avg"] = np.asarray(avg_dbw, dtype=np.float64)
    return out
