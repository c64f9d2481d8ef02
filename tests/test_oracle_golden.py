"""The oracle against every golden vector captured from the reference.

Bar: bit-exact on every Signal field (the oracle runs the same NumPy/SciPy
arithmetic as the reference), including NaN std for zero-power cells."""
import datetime

import numpy as np
import pytest

from oracle import analyze_oracle as oracle
from tests import golden_util as gu


def _assert_tables_identical(got, want):
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True), f"\n{got}\n!=\n{want}"


@pytest.mark.parametrize("name", gu.iq_case_names())
def test_oracle_matches_reference_on_iq_case(name):
    meta, kwargs, buffers, ts_starts, expected = gu.iq_case(name)
    oa = oracle.OracleAnalyzer(**{**dict(device="0"), **kwargs})
    for buf, ts, exp in zip(buffers, ts_starts, expected):
        every, kept = oa.process(buf, ts)
        ts_utc = ts.replace(tzinfo=datetime.timezone.utc)
        _assert_tables_identical(gu.signals_table(every, ts_utc), exp["table"])
        kept_ids = {id(s) for s in kept}
        assert [id(s) in kept_ids for s in every] == list(exp["kept"])
        assert oa.spec_last.shape == exp["spec_shape"]
        assert oa.spec_last.dtype == np.float32
        if "rows" in exp:
            assert np.array_equal(oa.spec_last[meta["keep_rows"], :], exp["rows"])
        means = np.array([np.mean(r) for r in oa.spec_last], dtype=np.float32)
        assert np.array_equal(means, exp["row_means"])


def test_kat1_known_answers():
    """the three pre-filter records of config 1."""
    meta, kwargs, buffers, ts_starts, expected = gu.iq_case("cfg1_tone")
    tab = expected[0]["table"]
    assert tab.shape == (3, 8)
    assert list(tab[:, 0]) == [299093.0] * 3 and list(tab[:, 2]) == [21333.0] * 3
    assert list(tab[:, 1]) == [150199218.75, 150200390.625, 150201562.5]
    assert tab[1, 3] == -72.80327606201172 and tab[1, 7] == 16.706167221069336
    assert list(expected[0]["kept"]) == [False, True, False]


@pytest.mark.parametrize("i", range(len(gu.extract_index())))
def test_oracle_extractor_on_planted_maps(i):
    c = gu.extract_case(i)
    p = oracle.ExtractParams(
        c["kwargs"]["signal_threshold_dbw"],
        c["kwargs"]["snr_threshold_db"],
        c["kwargs"]["signal_min_duration_ms"],
        c["kwargs"]["signal_max_duration_ms"],
        c["kwargs"]["calibration_db"],
    )
    last = c["last"] if c["has_last"] else None
    recs = oracle.extract_records(c["times"], c["cur"], last, p)
    sigs = oracle.records_to_signals(recs, c["freqs"], gu.TS0, "0", 150150000)
    _assert_tables_identical(gu.signals_table(sigs, gu.TS0_UTC), c["table"])
    kept_ids = {id(s) for s in oracle.filter_shadows(sigs)}
    assert [id(s) in kept_ids for s in sigs] == list(c["kept"])


def test_degenerate_lengths():
    p = oracle.ExtractParams()
    assert oracle.extract_records(np.zeros(0), np.zeros((4, 0), np.float32), None, p) == []
    with pytest.raises(IndexError):  # the reference indexes times[1]
        oracle.extract_records(np.array([0.1]), np.zeros((4, 1), np.float32), None, p)
