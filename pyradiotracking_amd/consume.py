"""Record serialisers: the reference's CSV / JSON / CBOR messages for arrays of records
(reference radiotracking/consume.py:23-55, 127-160, 165-199).

The formatting runs in the native library (include/rt_format.h, csrc/rt_format.cpp) on whole
arrays; this module is the host-side mirror of the reference interface:

* :func:`jsonify` / :func:`csvify` -- the value converters (consume.py:23-32, 50-55),
* :class:`CSVConsumer` -- same constructor and ``add(signal)`` as the reference class
  (consume.py:165-199), plus ``add_rows`` for record arrays,
* :func:`mqtt_messages` -- the (topic, payload) triples ``MQTTConsumer.add`` publishes for one
  message (consume.py:127-160); there is no MQTT client here (network I/O is out of scope),
* :func:`format_signals` / :func:`format_matched` -- the batch entries.

No Python fallback: without ``librt_analyze.so`` these raise.
"""
from __future__ import annotations

import ctypes as C
import datetime as _dt
from typing import Any, Iterable, List, Optional, Sequence, Tuple, Type

import numpy as np

import csv as _csv
import io as _io
import json as _json
import struct as _struct

from . import MatchedSignal, Signal, StateMessage
from . import _native
from .match import MatchedBatch, datetime_to_us

FORMAT_CSV, FORMAT_JSON, FORMAT_CBOR = 0, 1, 2
_KINDS = {"csv": FORMAT_CSV, "json": FORMAT_JSON, "cbor": FORMAT_CBOR}

FORMAT_SYMBOLS = ("rt_format_signals", "rt_format_matched", "rt_format_float_repr", "rt_signal_rows_from_records", "rt_records_keep_unshadowed",
                  "rt_host_set_threads")

# include/rt_format.h: rt_signal_row (72 B), rt_matched_row (24 B)
SIGNAL_ROW_DTYPE = np.dtype(
    [("device", "<i4"), ("reserved", "<i4"), ("ts_us", "<i8"), ("duration_us", "<i8"), ("frequency", "<f8"),
     ("max_dbw", "<f8"), ("avg_dbw", "<f8"), ("std_db", "<f8"), ("noise_dbw", "<f8"), ("snr_db", "<f8")]
)
MATCHED_ROW_DTYPE = np.dtype([("ts_us", "<i8"), ("duration_us", "<i8"), ("frequency", "<f8")])

_US = _dt.timedelta(microseconds=1)
_bound = None


def _lib():
    global _bound
    if _bound is not None:
        return _bound
    lib = _native.load_library()
    vp, sz = C.c_void_p, C.c_size_t
    lib.rt_format_signals.argtypes = [C.c_int32, vp, sz, vp, C.c_int32, vp, sz, vp, C.POINTER(sz)]
    lib.rt_format_matched.argtypes = [C.c_int32, vp, vp, vp, sz, vp, C.c_int32, vp, sz, vp, C.POINTER(sz)]
    lib.rt_format_float_repr.argtypes = [C.c_double, C.c_char_p]
    lib.rt_signal_rows_from_records.argtypes = [vp, sz, C.c_int32, C.c_double, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.rt_records_keep_unshadowed.restype = C.c_int
    lib.rt_records_keep_unshadowed.argtypes = [vp, sz, vp, C.POINTER(sz)]
    lib.rt_host_set_threads.argtypes = [C.c_int32]
    for name in FORMAT_SYMBOLS:
        getattr(lib, name)
    _bound = lib
    return lib


def set_host_threads(n: int = 0) -> int:
    """Threads the native sinks use per call (``rt_host_set_threads``): ``n`` > 0 exactly n, 0 = automatic (the machine's hardware
    threads, at most 32).  Process-wide; returns the number in force.  Output is byte-identical for any number."""
    return int(_lib().rt_host_set_threads(int(n)))


def jsonify(o):
    """``default=`` hook of the JSON messages: datetime -> ISO 8601, timedelta -> seconds."""
    if isinstance(o, _dt.datetime):
        return o.isoformat()
    if isinstance(o, _dt.timedelta):
        return o.total_seconds()
    raise TypeError(f"Object of type {type(o)} is not JSON serializable")


def csvify(o):
    """CSV cell converter: timedelta -> seconds, everything else unchanged."""
    return o.total_seconds() if isinstance(o, _dt.timedelta) else o


class Messages:
    """``n`` serialised messages back to back: ``data`` (bytes) and ``offsets`` (n + 1).  The native call's output buffer is kept
    as it is (``buffer``: a uint8 array, usable wherever a bytes-like object is -- ``file.write``, ``socket.send``); ``data`` makes
    the ``bytes`` object on first use (a copy of tens of megabytes for a few hundred thousand rows: not paid by consumers that
    write the buffer out or take single messages)."""

    __slots__ = ("buffer", "offsets", "_data")

    def __init__(self, data, offsets: np.ndarray):
        if isinstance(data, (bytes, bytearray)):
            self.buffer, self._data = np.frombuffer(data, dtype=np.uint8), bytes(data)
        else:
            self.buffer, self._data = data, None
        self.offsets = offsets

    @property
    def data(self) -> bytes:
        if self._data is None:
            self._data = self.buffer.tobytes()
        return self._data

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def __getitem__(self, i: int) -> bytes:
        return self.buffer[int(self.offsets[i]): int(self.offsets[i + 1])].tobytes()

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def _names(device_names: Sequence[str]):
    enc = [str(d).encode("utf-8") for d in device_names]
    arr = (C.c_char_p * max(1, len(enc)))(*enc)
    return arr, enc


def _call(fn, n: int, guess: int, *args) -> Messages:
    """One formatting pass into a buffer of ``guess`` bytes; only if that was too small a second one with
    the exact size the first pass reported."""
    offsets = np.zeros(n + 1, dtype=np.uintp)
    need = C.c_size_t(0)
    cap = max(1, guess)
    buf = np.empty(cap, dtype=np.uint8)  # (not ctypes.create_string_buffer: that one zero-fills -- 30 ms for a buffer 200 000 rows want)
    rc = fn(*args, buf.ctypes.data, cap, offsets.ctypes.data, C.byref(need))
    if rc == _native.RT_E_CAPACITY:
        cap = need.value
        buf = np.empty(max(1, cap), dtype=np.uint8)
        rc = fn(*args, buf.ctypes.data, cap, offsets.ctypes.data, C.byref(need))
    if rc != 0:
        raise _native.NativeError(rc, "rt_format: invalid arguments")
    return Messages(buf[: need.value], offsets)


def format_signals(kind: str, rows: np.ndarray, device_names: Sequence[str]) -> Messages:
    """``rows`` (SIGNAL_ROW_DTYPE) -> CSV rows (``\\r\\n`` terminated) / JSON documents / CBOR messages."""
    rows = np.ascontiguousarray(rows, dtype=SIGNAL_ROW_DTYPE)
    names, _keep = _names(device_names)
    guess = len(rows) * (320 + max((len(e) for e in _keep), default=0))
    return _call(_lib().rt_format_signals, len(rows), guess, _KINDS[kind], rows.ctypes.data, len(rows), names, len(device_names))


def format_matched(kind: str, rows: np.ndarray, avgs: np.ndarray, present: np.ndarray,
                   device_names: Sequence[str]) -> Messages:
    rows = np.ascontiguousarray(rows, dtype=MATCHED_ROW_DTYPE)
    nd = len(device_names)
    avgs = np.ascontiguousarray(avgs, dtype=np.float64).reshape(len(rows), nd)
    present = np.ascontiguousarray(present, dtype=np.uint8).reshape(len(rows), nd)
    names, _keep = _names(device_names)
    guess = len(rows) * (112 + sum(len(e) + 32 for e in _keep))
    return _call(_lib().rt_format_matched, len(rows), guess, _KINDS[kind], rows.ctypes.data, avgs.ctypes.data,
                 present.ctypes.data, len(rows), names, nd)


def format_matched_batch(kind: str, batch: MatchedBatch, device_names: Sequence[str]) -> Messages:
    """Groups as they come out of ``rt_match_add`` (match.MatchedBatch)."""
    rows = np.zeros(len(batch), dtype=MATCHED_ROW_DTYPE)
    for f in ("ts_us", "duration_us", "frequency"):
        rows[f] = batch.groups[f]
    return format_matched(kind, rows, batch.avgs, batch.present, device_names)


# ---------------------------------------------------------------------------
# records / objects -> rows
# ---------------------------------------------------------------------------
def signal_rows(signals: Iterable[Signal], device_names: List[str]) -> np.ndarray:
    """Signal objects -> SIGNAL_ROW_DTYPE; unknown device names are appended to ``device_names``."""
    signals = list(signals)
    index = {d: i for i, d in enumerate(device_names)}
    rows = np.zeros(len(signals), dtype=SIGNAL_ROW_DTYPE)
    for i, s in enumerate(signals):
        if s.device not in index:
            index[s.device] = len(device_names)
            device_names.append(s.device)
        rows[i] = (index[s.device], 0, datetime_to_us(s.ts), s.duration // _US, s.frequency, s.max, s.avg, s.std, s.noise, s.snr)
    return rows


def keep_unshadowed(rec: np.ndarray) -> np.ndarray:
    """``rec[rec["shadowed"] == 0]`` (what the reference hands to its consumers, analyze.py:248-251) on the host threads
    (``rt_records_keep_unshadowed``): NumPy's boolean mask over a million 40-byte records was a third of the records -> CSV path."""
    rec = np.ascontiguousarray(rec, dtype=_native.RECORD_DTYPE)
    if len(rec) < 4096:
        return np.ascontiguousarray(rec[rec["shadowed"] == 0])
    out = np.empty(len(rec), dtype=_native.RECORD_DTYPE)
    n = C.c_size_t(0)
    rc = _lib().rt_records_keep_unshadowed(rec.ctypes.data, len(rec), out.ctypes.data, C.byref(n))
    if rc != 0:
        raise _native.NativeError(rc, "rt_records_keep_unshadowed: invalid arguments")
    return out[: n.value]


def rows_from_analysis(rec: np.ndarray, decoder, ts_start_us: Sequence[int]) -> np.ndarray:
    """rt_record array of one analysis call -> SIGNAL_ROW_DTYPE (device = stream index), vectorised:
    the conversion of ``analyze._RecordDecoder`` without a Python object per signal.  Shadowed records
    are dropped (the reference never hands them to a consumer, analyze.py:248-251)."""
    r = keep_unshadowed(rec)
    rows = np.empty(len(r), dtype=SIGNAL_ROW_DTYPE)
    if not len(r):
        return rows
    # frequency and the five float32 dB figures by the decoder's NumPy expressions (the reference's own: their last digit is printed);
    # start time and duration from the cell coordinates in the library (rt_signal_rows_from_records: the reference's float64
    # expressions and timedelta's rounding, on the host threads)
    _t_start, _duration_s, frequency, max_dbw, avg_dbw, std_db, noise_dbw, snr_db = decoder.decode(r, times=False)
    cols = [np.ascontiguousarray(np.broadcast_to(c, len(r)), dtype=np.float32) for c in (max_dbw, avg_dbw, std_db, noise_dbw, snr_db)]
    frequency = np.ascontiguousarray(frequency, dtype=np.float64)
    ts0 = np.ascontiguousarray(ts_start_us, dtype=np.int64)
    rc = _lib().rt_signal_rows_from_records(r.ctypes.data, len(r), int(decoder.nperseg), float(decoder.sample_rate), ts0.ctypes.data, len(ts0),
                                            frequency.ctypes.data, *[c.ctypes.data for c in cols], rows.ctypes.data)
    if rc != 0:
        raise _native.NativeError(rc, "rt_signal_rows_from_records: invalid arguments (a record's stream outside ts_start_us?)")
    return rows


def _matched_arrays(msig: MatchedSignal):
    row = np.zeros(1, dtype=MATCHED_ROW_DTYPE)
    row[0] = (datetime_to_us(msig.ts), msig.duration // _US, msig.frequency)
    avgs = msig._avgs
    present = np.array([[a is not None for a in avgs]], dtype=np.uint8)
    vals = np.array([[a if a is not None else np.nan for a in avgs]], dtype=np.float64)
    return row, vals.reshape(1, len(avgs)), present.reshape(1, len(avgs))


def serialise(kind: str, message) -> bytes:
    """One ``Signal`` / ``MatchedSignal`` -> its CSV row (without line terminator), JSON document or CBOR message."""
    if isinstance(message, Signal):
        names = [message.device]
        out = format_signals(kind, signal_rows([message], names), names)[0]
    elif isinstance(message, MatchedSignal):
        row, vals, present = _matched_arrays(message)
        out = format_matched(kind, row, vals, present, message.devices[: vals.shape[1]])[0]
    elif isinstance(message, StateMessage):
        return _serialise_state(kind, message)
    else:
        raise TypeError(f"cannot serialise {type(message)}")
    return out[:-2] if kind == "csv" else out


def _serialise_state(kind: str, msg: StateMessage) -> bytes:
    """A StateMessage (one per analyzer and minute; as_list = [device, ts, state.value],
    __init__.py:84-90) through the standard library, CBOR by hand (array, text, tag 1, uint)."""
    if kind == "csv":
        buf = _io.StringIO()
        _csv.writer(buf, dialect="excel", delimiter=";").writerow([csvify(v) for v in msg.as_list])
        return buf.getvalue()[:-2].encode("utf-8")
    if kind == "json":
        return _json.dumps(msg.as_dict, default=jsonify).encode("ascii")
    if kind != "cbor":
        raise KeyError(kind)

    def head(major: int, v: int) -> bytes:
        if v < 24:
            return bytes([major << 5 | v])
        for info, fmt in ((24, ">B"), (25, ">H"), (26, ">I"), (27, ">Q")):
            if v < 1 << (8 * _struct.calcsize(fmt)):
                return bytes([major << 5 | info]) + _struct.pack(fmt, v)
        raise OverflowError(v)

    name = str(msg.device).encode("utf-8")
    sec, us = divmod(datetime_to_us(msg.ts), 10**6)
    if us:
        stamp = _struct.pack(">Bd", 0xFB, sec + us / 1000000)
    else:
        stamp = head(0, sec) if sec >= 0 else head(1, -1 - sec)
    return head(4, 3) + head(3, len(name)) + name + head(6, 1) + stamp + head(0, msg.state.value)


def mqtt_messages(message, prefix: str = "/radiotracking") -> List[Tuple[str, Any]]:
    """What ``MQTTConsumer.add`` publishes for one message (consume.py:127-160): topic
    ``<prefix>/device/<device>`` for a Signal, ``<prefix>/matched`` for a MatchingSignal, then
    ``/json`` (str), ``/csv`` (str, first line of the row) and ``/cbor`` (bytes)."""
    if isinstance(message, Signal):
        path = f"{prefix}/device/{message.device}"
    elif isinstance(message, MatchedSignal):
        path = f"{prefix}/matched"
    elif isinstance(message, StateMessage):
        path = f"{prefix}/state"
    else:
        return []
    csv_row = serialise("csv", message).decode("utf-8")
    return [
        (path + "/json", serialise("json", message).decode("ascii")),
        (path + "/csv", csv_row.splitlines()[0] if csv_row else csv_row),
        (path + "/cbor", serialise("cbor", message)),
    ]


class CSVConsumer:
    """Drop-in for ``radiotracking.consume.CSVConsumer`` (consume.py:165-199): writes the header once,
    then one ``;``-separated row per message of type ``cls``; other messages are ignored."""

    def __init__(self, out, cls: Type, header: Optional[List[str]] = None):
        self.out = out
        self.cls = cls
        if header:
            self.out.write(self._header_row(header))
        self.out.flush()

    @staticmethod
    def _header_row(header: Sequence[Any]) -> str:
        cells = []
        for h in header:
            h = "" if h is None else str(h)
            if any(c in h for c in ';"\r\n'):
                h = '"' + h.replace('"', '""') + '"'
            cells.append(h)
        return ";".join(cells) + "\r\n"

    def add(self, signal) -> None:
        if isinstance(signal, self.cls):
            self.out.write(serialise("csv", signal).decode("utf-8") + "\r\n")
            self.out.flush()

    def add_rows(self, rows: np.ndarray, device_names: Sequence[str]) -> int:
        """Batch entry for Signal rows (``rows_from_analysis`` / ``signal_rows``)."""
        msgs = format_signals("csv", rows, device_names)
        self.out.write(msgs.data.decode("utf-8"))
        self.out.flush()
        return len(msgs)

    def add_matched(self, batch: MatchedBatch, device_names: Sequence[str]) -> int:
        msgs = format_matched_batch("csv", batch, device_names)
        self.out.write(msgs.data.decode("utf-8"))
        self.out.flush()
        return len(msgs)
