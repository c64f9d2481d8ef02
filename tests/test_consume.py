"""Record serialisers: the native CSV / JSON formatter against payloads published
by the reference's own consumers (tests/golden/consume_cases.npz, made by make_golden_consume.py), the
drop-in CSVConsumer against the reference's file contents, and CBOR against RFC 8949 (cbor2 is not
installed here, so CBOR parity with the reference is unpinned; see the golden script's header).
Host code only: runs without a GPU."""
import csv
import datetime
import io
import json
import math
import os
import struct

import numpy as np
import pytest

from pyradiotracking_amd import MatchedSignal, MatchingSignal, Signal, build
from pyradiotracking_amd import consume as rtc
from pyradiotracking_amd.match import us_to_datetime

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "consume_cases.npz"))
US = datetime.timedelta(microseconds=1)


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_library()


def signal_rows():
    names = [str(x) for x in G["devices"]]
    rows = np.zeros(len(G["sig_ts_us"]), dtype=rtc.SIGNAL_ROW_DTYPE)
    rows["device"] = [names.index(str(d)) for d in G["sig_device"]]
    rows["ts_us"], rows["duration_us"], rows["frequency"] = G["sig_ts_us"], G["sig_dur_us"], G["sig_freq"]
    for k, f in enumerate(("max_dbw", "avg_dbw", "std_db", "noise_dbw", "snr_db")):
        rows[f] = G["sig_vals"][:, k]
    return rows, names


def matched_rows():
    rows = np.zeros(len(G["m_ts_us"]), dtype=rtc.MATCHED_ROW_DTYPE)
    rows["ts_us"], rows["duration_us"], rows["frequency"] = G["m_ts_us"], G["m_dur_us"], G["m_freq"]
    return rows, G["m_avgs"], G["m_present"], [str(x) for x in G["m_devices"]]


def test_signal_json_and_csv_equal_the_reference_payloads():
    rows, names = signal_rows()
    js = rtc.format_signals("json", rows, names)
    cs = rtc.format_signals("csv", rows, names)
    assert len(js) == len(cs) == len(rows) == 260
    for i in range(len(rows)):
        assert js[i].decode("ascii") == str(G["sig_json"][i]), i
        row = cs[i].decode("utf-8")
        assert row.endswith("\r\n")
        # MQTT publishes `getvalue().splitlines()[0]` (consume.py:150)
        assert row[:-2].splitlines()[0] == str(G["sig_csv"][i]), i


def test_matched_json_and_csv_equal_the_reference_payloads():
    rows, avgs, present, names = matched_rows()
    js = rtc.format_matched("json", rows, avgs, present, names)
    cs = rtc.format_matched("csv", rows, avgs, present, names)
    for i in range(len(rows)):
        assert js[i].decode("ascii") == str(G["m_json"][i]), i
        assert cs[i].decode("utf-8")[:-2] == str(G["m_csv"][i]), i


def _signals():
    rows, names = signal_rows()
    return [Signal(names[r["device"]], us_to_datetime(r["ts_us"]), r["frequency"], int(r["duration_us"]) * US, r["max_dbw"],
                   r["avg_dbw"], r["std_db"], r["noise_dbw"], r["snr_db"]) for r in rows]


def test_csv_consumer_writes_the_reference_file():
    out = io.StringIO()
    c = rtc.CSVConsumer(out, cls=Signal, header=Signal.header)
    for s in _signals():
        c.add(s)
        c.add("not a signal")  # consume.py:194: other types are ignored
    assert out.getvalue() == str(G["sig_csv_file"])
    # batch entry: the same file from arrays
    out2 = io.StringIO()
    c2 = rtc.CSVConsumer(out2, cls=Signal, header=Signal.header)
    rows, names = signal_rows()
    assert c2.add_rows(rows, names) == len(rows)
    assert out2.getvalue() == str(G["sig_csv_file"])
    # what was written parses back with the csv module to the as_list values
    back = list(csv.reader(io.StringIO(out.getvalue(), newline=""), dialect="excel", delimiter=";"))
    assert back[0] == Signal.header and len(back) == 1 + len(rows)
    assert back[1][0] == names[rows["device"][0]] and float(back[1][2]) == rows["frequency"][0]


def _matched_objects():
    rows, avgs, present, names = matched_rows()
    out = []
    for r, a, p in zip(rows, avgs, present):
        out.append(MatchingSignal.from_aggregate(names, us_to_datetime(r["ts_us"]), r["frequency"], int(r["duration_us"]) * US,
                                                 [float(x) if q else None for x, q in zip(a, p)]))
    return out, names


def test_matched_csv_consumer_and_mqtt_messages():
    groups, names = _matched_objects()
    out = io.StringIO()
    c = rtc.CSVConsumer(out, cls=MatchingSignal, header=MatchingSignal(names).header)
    for g in groups:
        c.add(g)
    c.add(_signals()[0])  # a Signal is not a MatchingSignal
    assert out.getvalue() == str(G["m_csv_file"])
    for i, g in enumerate(groups):
        msgs = rtc.mqtt_messages(g, prefix="station/radiotracking")
        assert [t for t, _ in msgs] == [str(t) for t in G["m_topics"][i]]
        assert msgs[0][1] == str(G["m_json"][i]) and msgs[1][1] == str(G["m_csv"][i])
    sigs = _signals()
    for i in (0, 1, 7, 203, 215, 231, 259):
        msgs = rtc.mqtt_messages(sigs[i], prefix="station/radiotracking")
        assert [t for t, _ in msgs] == [str(t) for t in G["sig_topics"][i]]
        assert msgs[0][1] == str(G["sig_json"][i]) and msgs[1][1] == str(G["sig_csv"][i])
        assert isinstance(msgs[2][1], bytes) and msgs[2][1][0] == 0x89
    assert rtc.mqtt_messages("something else") == []


def test_value_converters():
    assert rtc.csvify(datetime.timedelta(milliseconds=20)) == 0.02 and rtc.csvify(5) == 5
    assert rtc.jsonify(datetime.timedelta(seconds=2)) == 2.0
    assert rtc.jsonify(datetime.datetime(2024, 1, 1, tzinfo=datetime.timezone.utc)) == "2024-01-01T00:00:00+00:00"
    with pytest.raises(TypeError):
        rtc.jsonify(object())
    # json.dumps(as_dict, default=jsonify) of the message types equals the native document
    s = _signals()[3]
    assert json.dumps(s.as_dict, default=rtc.jsonify) == rtc.serialise("json", s).decode()
    g = _matched_objects()[0][5]
    assert json.dumps(g.as_dict, default=rtc.jsonify) == rtc.serialise("json", g).decode()


def test_json_and_csv_of_every_ascii_character_in_a_device_name():
    """json.dumps (ensure_ascii) escapes everything outside ' '..'~' -- control characters AND DEL (0x7f) -- and
    csv quotes on the delimiter, the quote character and line breaks; the native formatters must agree with the
    standard library the reference uses for each of the 127 characters (0x7f was found by tests/perf/soak_consume.py)."""
    import csv
    import io

    from pyradiotracking_amd import Signal

    ts = datetime.datetime(2024, 1, 1, 0, 0, 1, 5, tzinfo=datetime.timezone.utc)
    for code in list(range(1, 128)) + [0xE4, 0x2028, 0x1F4E1]:
        name = "a" + chr(code) + "b"
        s = Signal(name, ts, 150.1e6, datetime.timedelta(milliseconds=20), -70.5, -72.25, 1.5, -100.0, 20.0)
        assert rtc.serialise("json", s).decode("ascii") == json.dumps(s.as_dict, default=rtc.jsonify), hex(code)
        buf = io.StringIO()
        csv.writer(buf, dialect="excel", delimiter=";").writerow([rtc.csvify(v) for v in s.as_list])
        assert buf.getvalue().endswith("\r\n")
        assert rtc.serialise("csv", s).decode("utf-8") == buf.getvalue()[:-2], hex(code)  # the row without its terminator


# ---------------------------------------------------------------------------
# CBOR: decoded with a minimal RFC 8949 reader written for this test
# ---------------------------------------------------------------------------
def cbor_item(b, i=0):
    ib = b[i]
    major, info = ib >> 5, ib & 31
    i += 1
    if major == 7:
        if info == 22:
            return None, i
        if info == 25:
            return struct.unpack(">e", b[i:i + 2])[0], i + 2
        if info == 27:
            return struct.unpack(">d", b[i:i + 8])[0], i + 8
        raise AssertionError(f"unexpected simple/float {info}")
    if info < 24:
        val = info
    else:
        n = {24: 1, 25: 2, 26: 4, 27: 8}[info]
        val = int.from_bytes(b[i:i + n], "big")
        i += n
    if major == 0:
        return val, i
    if major == 1:
        return -1 - val, i
    if major == 3:
        return b[i:i + val].decode("utf-8"), i + val
    if major == 4:
        items = []
        for _ in range(val):
            x, i = cbor_item(b, i)
            items.append(x)
        return items, i
    if major == 6:
        x, i = cbor_item(b, i)
        return ("tag", val, x), i
    raise AssertionError(f"unexpected major type {major}")


def same_float(a, b):
    return (math.isnan(a) and math.isnan(b)) or (a == b and math.copysign(1, a) == math.copysign(1, b))


def test_cbor_messages_decode_to_as_list():
    rows, names = signal_rows()
    msgs = rtc.format_signals("cbor", rows, names)
    for i, r in enumerate(rows):
        item, end = cbor_item(msgs[i])
        assert end == len(msgs[i]) and len(item) == 9
        assert item[0] == names[r["device"]]
        tag, num, ts = item[1]
        assert (tag, num) == ("tag", 1)
        sec, us = divmod(int(r["ts_us"]), 10**6)
        if us == 0:
            assert isinstance(ts, int) and ts == sec  # cbor2: an int when microsecond == 0
        else:
            assert isinstance(ts, float) and ts == sec + us / 1000000
        assert same_float(item[2], r["frequency"])
        assert item[3][:2] == ("tag", 1337) and item[3][2] == int(r["duration_us"]) / 10**6
        for k, f in enumerate(("max_dbw", "avg_dbw", "std_db", "noise_dbw", "snr_db")):
            assert same_float(item[4 + k], r[f])
    mrows, avgs, present, mnames = matched_rows()
    mm = rtc.format_matched("cbor", mrows, avgs, present, mnames)
    for i, r in enumerate(mrows):
        item, end = cbor_item(mm[i])
        assert end == len(mm[i]) and len(item) == 3 + len(mnames)
        assert item[3:] == [float(a) if p else None for a, p in zip(avgs[i], present[i])]


def test_cbor_known_answers_of_rfc8949():
    """Appendix A of RFC 8949: 1.1 -> fb3ff199999999999a, epoch 1363896240 -> c11a514b67b0,
    1363896240.5 -> c1fb41d452d9ec200000, NaN -> f97e00, Infinity -> f97c00, -Infinity -> f9fc00, null -> f6."""
    rows = np.zeros(2, dtype=rtc.SIGNAL_ROW_DTYPE)
    rows["ts_us"] = [1363896240 * 10**6, 1363896240 * 10**6 + 500000]
    rows["duration_us"] = [1100000, 0]
    rows["frequency"] = [1.1, 100000.0]
    rows["max_dbw"] = [np.nan, 1.0e300]
    rows["avg_dbw"] = [np.inf, -4.1]
    rows["std_db"] = [-np.inf, 0.0]
    rows["noise_dbw"] = [-0.0, 5.960464477539063e-8]
    rows["snr_db"] = [1.5, 3.4028234663852886e38]
    msgs = rtc.format_signals("cbor", rows, ["a"])
    assert msgs[0].hex() == ("89" "6161" "c11a514b67b0" "fb3ff199999999999a" "d90539fb3ff199999999999a" "f97e00" "f97c00" "f9fc00"
                             "fb8000000000000000" "fb3ff8000000000000")
    assert msgs[1].hex() == ("89" "6161" "c1fb41d452d9ec200000" "fb40f86a0000000000" "d90539fb0000000000000000" "fb7e37e43c8800759c"
                             "fbc010666666666666" "fb0000000000000000" "fb3e70000000000000" "fb47efffffe0000000")
    m = rtc.format_matched("cbor", np.array([(10**6, 2 * 10**6, 1.1)], dtype=rtc.MATCHED_ROW_DTYPE), np.array([[1.5, np.nan]]),
                           np.array([[1, 0]], dtype=np.uint8), ["x", "y"])
    assert m[0].hex() == "85" "c101" "fb3ff199999999999a" "d90539fb4000000000000000" "fb3ff8000000000000" "f6"
    wide = rtc.format_matched("cbor", np.zeros(1, dtype=rtc.MATCHED_ROW_DTYPE), np.zeros((1, 30)), np.zeros((1, 30), dtype=np.uint8),
                              [str(i) for i in range(30)])
    assert wide[0][:2].hex() == "9821"  # array of 33 items: one-byte length
    assert wide[0][2:].hex() == "c100" "fb0000000000000000" "d90539fb0000000000000000" + "f6" * 30


def _libcbor():
    import ctypes as C
    import ctypes.util

    path = ctypes.util.find_library("cbor") or "libcbor.so.0.8"
    try:
        lib = C.CDLL(path)
    except OSError:
        return None

    class LoadResult(C.Structure):
        _fields_ = [("error_position", C.c_size_t), ("error_code", C.c_int), ("read", C.c_size_t)]

    lib.cbor_load.restype = C.c_void_p
    lib.cbor_load.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(LoadResult)]
    lib.cbor_serialize_alloc.restype = C.c_size_t
    lib.cbor_serialize_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.cbor_decref.argtypes = [C.POINTER(C.c_void_p)]
    lib.cbor_isa_array.restype = C.c_bool
    lib.cbor_isa_array.argtypes = [C.c_void_p]
    lib.cbor_array_size.restype = C.c_size_t
    lib.cbor_array_size.argtypes = [C.c_void_p]
    return lib, LoadResult


def test_cbor_messages_pass_through_libcbor_unchanged():
    """A second RFC 8949 implementation (libcbor 0.8, the C library of the image; not cbor2, so the parity with the
    reference's encoder stays unpinned): every message loads without error, is consumed to its last byte, is one
    array of the expected length, and libcbor's own serialisation of the loaded item gives the same bytes back."""
    import ctypes as C

    got = _libcbor()
    if got is None:
        pytest.skip("libcbor is not installed")
    lib, LoadResult = got
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rows, names = signal_rows()
    mrows, avgs, present, mnames = matched_rows()
    batches = [(rtc.format_signals("cbor", rows, names), 9), (rtc.format_matched("cbor", mrows, avgs, present, mnames), 3 + len(mnames))]
    n = 0
    for msgs, width in batches:
        for m in msgs:
            raw = bytes(m)
            res = LoadResult()
            item = lib.cbor_load(raw, len(raw), C.byref(res))
            assert item and res.error_code == 0 and res.read == len(raw), (res.error_code, res.error_position, raw.hex())
            assert lib.cbor_isa_array(item) and lib.cbor_array_size(item) == width
            buf, size = C.c_void_p(), C.c_size_t()
            length = lib.cbor_serialize_alloc(item, C.byref(buf), C.byref(size))
            assert length == len(raw) and C.string_at(buf, length) == raw
            libc.free(buf)
            ref = C.c_void_p(item)
            lib.cbor_decref(C.byref(ref))
            n += 1
    assert n == len(rows) + len(mrows) and n > 100


def test_errors_and_empty_batches():
    from pyradiotracking_amd import _native

    assert len(rtc.format_signals("csv", np.zeros(0, dtype=rtc.SIGNAL_ROW_DTYPE), ["0"])) == 0
    bad = np.zeros(1, dtype=rtc.SIGNAL_ROW_DTYPE)
    bad["device"] = 3
    with pytest.raises(_native.NativeError):
        rtc.format_signals("csv", bad, ["0"])
    with pytest.raises(KeyError):
        rtc.format_signals("xml", bad, ["0"])
    with pytest.raises(TypeError):
        rtc.serialise("csv", 5)


def test_state_messages_format_like_the_reference():
    from pyradiotracking_amd import StateMessage

    out = io.StringIO()
    c = rtc.CSVConsumer(out, cls=StateMessage, header=StateMessage.header)
    for i in range(len(G["st_ts_us"])):
        m = StateMessage(str(G["st_device"][i]), us_to_datetime(G["st_ts_us"][i]), int(G["st_state"][i]))
        assert repr(m) == str(G["st_repr"][i])
        assert m.as_dict == {"Device": m.device, "Time": m.ts, "State": m.state.value}
        msgs = rtc.mqtt_messages(m, prefix="station/radiotracking")
        assert [t for t, _ in msgs] == [str(t) for t in G["st_topics"][i]]
        assert msgs[0][1] == str(G["st_json"][i]) and msgs[1][1] == str(G["st_csv"][i])
        item, end = cbor_item(msgs[2][1])
        assert end == len(msgs[2][1]) and item[0] == m.device and item[2] == m.state.value
        sec, us = divmod(int(G["st_ts_us"][i]), 10**6)
        assert item[1] == ("tag", 1, sec if us == 0 else sec + us / 1000000)
        c.add(m)
    c.add(_signals()[0])
    assert out.getvalue() == str(G["st_csv_file"])
    assert StateMessage("x", us_to_datetime(0), StateMessage.State.RUNNING).state is StateMessage.State(1)
    assert StateMessage("x", us_to_datetime(0), "2").state is StateMessage.State.STARTED


@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 70001])
@pytest.mark.parametrize("threads", [1, 5])
def test_keep_unshadowed_is_the_boolean_mask(n, threads):
    """``consume.keep_unshadowed`` (``rt_records_keep_unshadowed``: blocks count, a prefix places them, blocks copy -- on the host
    threads) returns ``rec[rec["shadowed"] == 0]``, the records the reference hands to its consumers (analyze.py:248-251)."""
    from pyradiotracking_amd import _native

    rng = np.random.default_rng(n + threads)
    rec = np.zeros(n, dtype=_native.RECORD_DTYPE)
    rec["stream"] = rng.integers(0, 9, n)
    rec["fi"] = rng.integers(0, 256, n)
    rec["start"] = rng.integers(-5, 900, n)
    rec["end"] = rec["start"] + rng.integers(1, 50, n)
    rec["max_p"] = rng.uniform(1e-9, 1e-6, n).astype(np.float32)
    rec["shadowed"] = rng.random(n) < 0.6
    rtc.set_host_threads(threads)
    try:
        got = rtc.keep_unshadowed(rec)
    finally:
        rtc.set_host_threads(0)
    want = rec[rec["shadowed"] == 0]
    assert got.dtype == want.dtype and got.tobytes() == want.tobytes()


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_threaded_sinks_are_byte_identical_to_one_thread(threads):
    """The native sinks deal blocks of 2 048 rows to `set_host_threads` threads and assemble the result in block order: CSV / JSON /
    CBOR bytes, message offsets and the rows built from analysis records are the same for any number of threads -- and the rows are
    what the per-signal Python conversion gives (timedelta rounding included)."""
    import datetime

    from pyradiotracking_amd import _native
    from pyradiotracking_amd.analyze import _RecordDecoder
    from pyradiotracking_amd.match import datetime_to_us

    rng = np.random.default_rng(9)
    n, S = 7001, 37  # (three blocks and a bit)
    rec = np.zeros(n, dtype=_native.RECORD_DTYPE)
    rec["stream"] = np.sort(rng.integers(0, S, n))
    rec["fi"] = rng.integers(0, 256, n)
    rec["start"] = rng.integers(-40, 1100, n)
    rec["end"] = np.maximum(rec["start"], 0) + rng.integers(1, 60, n)
    rec["max_p"] = rng.uniform(1e-10, 1e-6, n)
    rec["mean_p"] = rec["max_p"] * rng.uniform(0.3, 1.0, n)
    rec["row_mean"] = rng.uniform(1e-13, 1e-11, n)
    rec["std_db"] = rng.uniform(0, 25, n)
    rec["shadowed"] = rng.random(n) < 0.2
    dec = _RecordDecoder(256, 300000, 150150000, list(rng.uniform(-3, 3, S)))
    names = [f"sdr{i}" for i in range(S)]
    ts0 = [datetime_to_us(datetime.datetime(2024, 1, 1, tzinfo=datetime.timezone.utc)) + 1000003 * i for i in range(S)]

    def python_rows():
        r = rec[rec["shadowed"] == 0]
        t_start, duration_s, frequency, max_dbw, avg_dbw, std_db, noise_dbw, snr_db = dec.decode(r)
        us = datetime.timedelta(microseconds=1)
        rows = np.zeros(len(r), dtype=rtc.SIGNAL_ROW_DTYPE)
        rows["device"] = r["stream"]
        rows["ts_us"] = np.asarray(ts0)[r["stream"]] + np.array([datetime.timedelta(seconds=float(v)) // us for v in t_start])
        rows["duration_us"] = [datetime.timedelta(seconds=float(v)) // us for v in duration_s]
        rows["frequency"] = frequency
        for name, col in (("max_dbw", max_dbw), ("avg_dbw", avg_dbw), ("std_db", std_db), ("noise_dbw", noise_dbw), ("snr_db", snr_db)):
            rows[name] = np.asarray(col, dtype=np.float64)
        return rows

    want_rows = python_rows()
    rtc.set_host_threads(1)
    one = {k: rtc.format_signals(k, want_rows, names) for k in ("csv", "json", "cbor")}
    try:
        assert rtc.set_host_threads(threads) == threads
        rows = rtc.rows_from_analysis(rec, dec, ts0)
        assert rows.tobytes() == want_rows.tobytes()
        for k in ("csv", "json", "cbor"):
            m = rtc.format_signals(k, rows, names)
            assert m.data == one[k].data and np.array_equal(m.offsets, one[k].offsets), k
        # the matched-signal messages alike
        mr = np.zeros(5000, dtype=rtc.MATCHED_ROW_DTYPE)
        mr["ts_us"], mr["duration_us"], mr["frequency"] = rows["ts_us"][:5000], rows["duration_us"][:5000], rows["frequency"][:5000]
        avgs = rng.uniform(-90, -40, (5000, 4))
        present = (rng.random((5000, 4)) < 0.7).astype(np.uint8)
        rtc.set_host_threads(1)
        ref = {k: rtc.format_matched(k, mr, avgs, present, names[:4]) for k in ("csv", "json", "cbor")}
        rtc.set_host_threads(threads)
        for k in ("csv", "json", "cbor"):
            m = rtc.format_matched(k, mr, avgs, present, names[:4])
            assert m.data == ref[k].data and np.array_equal(m.offsets, ref[k].offsets), k
    finally:
        rtc.set_host_threads(0)
    assert len(one["csv"]) == len(want_rows) and one["csv"][0].endswith(b"\r\n")
