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
