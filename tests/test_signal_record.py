"""The `Signal` record type (reference radiotracking/__init__.py:110-202) against what the reference printed:
``repr(Signal)`` strings captured by tests/golden/make_golden.py from the imported reference for every
pre-filter signal of every golden IQ case."""
import datetime

import numpy as np
import pytest

from pyradiotracking_amd import Signal
from tests import golden_util as gu


@pytest.mark.parametrize("name", gu.iq_case_names())
def test_repr_of_signals_built_from_the_reference_field_values(name):
    """Signal objects constructed from the stored field values print exactly what the reference's objects printed."""
    meta, kwargs, buffers, ts_starts, expected = gu.iq_case(name)
    n = 0
    for ts, exp in zip(ts_starts, expected):
        ts_utc = ts.replace(tzinfo=datetime.timezone.utc)
        assert len(exp["reprs"]) == len(exp["table"])
        for row, want in zip(exp["table"], exp["reprs"]):
            sig = Signal(
                kwargs.get("device", "0"),
                ts_utc + datetime.timedelta(microseconds=int(row[0])),
                row[1],
                datetime.timedelta(microseconds=int(row[2])),
                np.float32(row[3]), np.float32(row[4]), np.float32(row[5]), np.float32(row[6]), np.float32(row[7]),
            )
            assert repr(sig) == want
            n += 1
    assert n == sum(len(e["reprs"]) for e in expected)


def test_repr_and_str_formats():
    s = Signal("7", "2024-01-01T00:00:00.299093+00:00", 150200390.625, 0.021333, -72.80327606201172, -73.1, 17.0, -89.8, 16.7)
    assert repr(s) == "Signal(7, 2024-01-01 00:00:00.299093+00:00, 150200390.625, 0:00:00.021333, -72.80327606201172, -73.1, 17.0, -89.8, 16.7)"
    assert str(s) == "Signal<SDR 7, 150.200 MHz, 21.33 ms, -72.8 dBW>"


def test_column_wise_signals_equal_the_record_by_record_construction():
    """`_RecordDecoder.signals` builds the nine fields column-wise and fills the objects slot by slot; the result must be what
    the reference's per-record expressions give (analyze.py:420-450) -- also across a change of the zone's UTC offset inside
    the buffer, where the per-stream shortcut for `astimezone` does not apply -- and `signal_batch` the same, lazily."""
    import datetime

    import numpy as np
    import pytz

    from pyradiotracking_amd import Signal, _native
    from pyradiotracking_amd.analyze import SignalBatch, _RecordDecoder

    rng = np.random.default_rng(3)
    n, n_streams = 4000, 7
    rec = np.zeros(n, dtype=_native.RECORD_DTYPE)
    rec["stream"] = np.sort(rng.integers(0, n_streams, n))
    rec["fi"] = rng.integers(0, 256, n)
    rec["start"] = rng.integers(-40, 7900, n)
    rec["end"] = np.maximum(rec["start"], 0) + rng.integers(2, 130, n)
    rec["max_p"] = rng.uniform(1e-9, 1e-6, n)
    rec["mean_p"] = rec["max_p"] * 0.7
    rec["std_db"] = rng.uniform(1, 20, n)
    rec["row_mean"] = 1e-12
    rec["max_p"][5] = 0.0  # -inf dBW
    dec = _RecordDecoder(256, 2048000, 150150000, [0.5 * s for s in range(n_streams)])
    names = [f"sdr{i}" for i in range(n_streams)]
    berlin = pytz.timezone("Europe/Berlin")
    for ts_starts in ([datetime.datetime(2024, 1, 1, 12, 0, s) for s in range(n_streams)],
                      [berlin.localize(datetime.datetime(2024, 3, 31, 1, 59, 57)) for _ in range(n_streams)],          # +01:00, the offset is a fixed one
                      [datetime.datetime(2024, 3, 31, 1, 59, 57, tzinfo=berlin) for _ in range(n_streams)]):           # pytz's LMT quirk: still one offset
        t_start, duration_s, frequency, max_dbw, avg_dbw, std_db, noise_dbw, snr_db = dec.decode(rec)
        want = []
        for i in range(n):
            s = int(rec["stream"][i])
            ts = ts_starts[s] + datetime.timedelta(seconds=float(t_start[i]))
            want.append(Signal(names[s], ts.astimezone(pytz.utc), frequency[i], datetime.timedelta(seconds=float(duration_s[i])),
                               max_dbw[i], avg_dbw[i], std_db[i], noise_dbw[i], snr_db[i]))
        got = dec.signals(rec, names, ts_starts)
        assert [repr(g) for g in got] == [repr(w) for w in want]
        assert all(type(g.max) is float and type(g.frequency) is float and type(g.duration) is datetime.timedelta for g in got)
        lazy = dec.signal_batch(rec, names, ts_starts)
        assert isinstance(lazy, SignalBatch) and len(lazy) == n and repr(lazy[17]) == repr(want[17]) and [repr(x) for x in lazy[100:103]] == [repr(w) for w in want[100:103]]
    assert dec.signals(rec[:0], names, ts_starts) == [] and len(dec.signal_batch(rec[:0], names, ts_starts)) == 0

    class Flip(datetime.tzinfo):  # a zone whose offset changes one second into the buffer: the shortcut must not be taken
        def utcoffset(self, dt):
            return datetime.timedelta(hours=1 if dt.replace(tzinfo=None) < datetime.datetime(2024, 1, 1, 0, 0, 1) else 2)

        def dst(self, dt):
            return datetime.timedelta(0)

    ts_starts = [datetime.datetime(2024, 1, 1, 0, 0, 0, tzinfo=Flip())] * n_streams
    t_start = dec.decode(rec)[0]
    got = dec.signals(rec, names, ts_starts)
    for i in (0, 1, n // 2, n - 1):
        assert got[i].ts == (ts_starts[0] + datetime.timedelta(seconds=float(t_start[i]))).astimezone(pytz.utc)
