"""CPU tests of the scalar decision logic the detect kernels execute.

csrc/rt_core.h is compiled for the host (_rt_hostcheck.so, tests only) and
driven against the oracle / golden vectors: predicate, strided-probe rule,
start walk with look-back, float64 duration gate, statistics, CPython
timedelta rounding, ordering and shadow verdicts.  No GPU, no FFT."""
import ctypes as C
import datetime
import os

import numpy as np
import pytest

from oracle import analyze_oracle as oracle
from pyradiotracking_amd import _native, build
from tests import golden_util as gu


@pytest.fixture(scope="module")
def hc():
    path = build.build_hostcheck()
    lib = C.CDLL(path)
    lib.hc_timedelta_us.argtypes = [C.c_double]
    lib.hc_timedelta_us.restype = C.c_longlong
    lib.hc_probe_stride.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.hc_seg_time.argtypes = [C.c_int, C.c_int, C.c_double]
    lib.hc_seg_time.restype = C.c_double
    lib.hc_tail_cols.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.hc_extract.argtypes = [
        C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
        C.c_float, C.c_float, C.c_float, C.c_double, C.c_double, C.c_void_p, C.c_int,
    ]
    return lib


def _us(td):
    return td.days * 86400 * 10**6 + td.seconds * 10**6 + td.microseconds


def test_timedelta_rounding_matches_cpython(hc):
    rng = np.random.default_rng(3)
    vals = list(rng.uniform(-2, 2, 4000)) + list(rng.uniform(-1e-3, 1e-3, 2000))
    vals += [k * 0.5e-6 for k in range(-50, 50)] + [0.0, 1.0, -1.0, 0.0213335, 0.2990935, -0.0106665]
    vals += [(k + 0.5) * 1e-6 for k in range(-20, 20)] + [1 + (k + 0.5) * 1e-6 for k in range(-20, 20)]
    for v in vals:
        assert hc.hc_timedelta_us(float(v)) == _us(datetime.timedelta(seconds=float(v))), v


@pytest.mark.parametrize("nperseg,fs", [(256, 300000), (256, 2048000), (1024, 2400000), (4096, 3200000), (256, 1000000)])
def test_time_axis_and_stride(hc, nperseg, fs):
    n = 50 * nperseg + 17
    times = np.arange(nperseg / 2, n - nperseg / 2 + 1, nperseg) / float(fs)
    for k in (0, 1, 2, 7, 49):
        assert hc.hc_seg_time(k, nperseg, float(fs)) == times[k]
    for min_ms in (8, 5, 1, 0.01, 8.0000001, 7.9999):
        want = max(1, int((min_ms / 1000) / (times[1] - times[0])))
        assert hc.hc_probe_stride(nperseg, float(fs), min_ms / 1000) == want


def _run_hostcheck(hc, c, tail_cols=None):
    kw = c["kwargs"]
    p = oracle.ExtractParams(
        kw["signal_threshold_dbw"], kw["snr_threshold_db"], kw["signal_min_duration_ms"],
        kw["signal_max_duration_ms"], kw["calibration_db"],
    )
    cur = np.ascontiguousarray(c["cur"].T, dtype=np.float32)
    n_seg, n_bins = cur.shape
    last = np.ascontiguousarray(c["last"].T, dtype=np.float32) if c["has_last"] else None
    n_last = last.shape[0] if last is not None else 0
    out = np.zeros(512, dtype=_native.RECORD_DTYPE)
    n = hc.hc_extract(
        cur.ctypes.data, n_seg, n_bins, last.ctypes.data if last is not None and last.size else (C.c_void_p(8) if last is not None else None),
        n_last, n_last if tail_cols is None else min(tail_cols, n_last), 256, float(kw["sample_rate"]),
        np.float32(p.signal_threshold), np.float32(p.snr_threshold), np.float32(kw["calibration_db"]),
        p.signal_min_duration, p.signal_max_duration, out.ctypes.data, len(out),
    )
    return out[:n], p


def _compare_records(rec, want_records, want_kept):
    assert len(rec) == len(want_records)
    for r, w, k in zip(rec, want_records, want_kept):
        assert (r["fi"], r["start"], r["end"]) == (w.fi, w.start, w.end)
        assert bool(r["shadowed"]) == (not k)
        for got_lin, want_db, cal in ((r["max_p"], w.max_dbw, True), (r["mean_p"], w.avg_dbw, True), (r["row_mean"], w.noise_dbw, False)):
            got_db = 10 * np.log10(np.float32(got_lin))
            if np.isfinite(want_db + 0.0):
                assert abs(got_db - (want_db + (cal and c_cal[0] or 0.0))) < 1e-4
        if np.isnan(w.std_db):
            assert np.isnan(r["std_db"])
        else:
            assert abs(r["std_db"] - w.std_db) < 1e-3


c_cal = [0.0]


@pytest.mark.parametrize("i", range(len(gu.extract_index())))
def test_core_logic_on_planted_maps(hc, i):
    c = gu.extract_case(i)
    rec, p = _run_hostcheck(hc, c)
    last = c["last"] if c["has_last"] else None
    want = oracle.extract_records(c["times"], c["cur"], last, p)
    sigs = oracle.records_to_signals(want, c["freqs"], gu.TS0, "0", 150150000)
    kept_ids = {id(s) for s in oracle.filter_shadows(sigs)}
    c_cal[0] = c["kwargs"]["calibration_db"]
    _compare_records(rec, want, [id(s) in kept_ids for s in sigs])
    # and against the golden table captured from the reference itself
    assert len(rec) == len(c["table"])
    assert [not bool(r["shadowed"]) for r in rec] == list(c["kept"])


@pytest.mark.parametrize("i", range(0, len(gu.extract_index()), 3))
def test_bounded_tail_is_exact(hc, i):
    """Reading only floor(max_d/hop)+2 trailing columns of the previous map
    gives the same records as unlimited look-back (Appendix A.2 'tail sufficiency')."""
    c = gu.extract_case(i)
    full, p = _run_hostcheck(hc, c)
    k = hc.hc_tail_cols(256, float(c["kwargs"]["sample_rate"]), p.signal_max_duration)
    cut, _ = _run_hostcheck(hc, c, tail_cols=k)
    assert full.tobytes() == cut.tobytes()


def test_auto_levels_form_a_ladder(hc):
    """RT_MODE_AUTO's level bookkeeping (rt_core.h: level_up / level_down / level_rank): sparse < chunk bits < exact <
    dense, the middle levels only where they exist; up and down are inverse on the levels that exist."""
    D, S, P, R = _native.RT_MODE_DENSE, _native.RT_MODE_SPARSE, _native.RT_MODE_PREFILTER, _native.RT_MODE_RUNFILTER
    assert [hc.hc_level_rank(m) for m in (S, P, R, D)] == [0, 1, 2, 3]
    for pre in (0, 1):
        for run in (0, 1):
            ladder = [S] + ([P] if pre else []) + ([R] if run else []) + [D]
            for lo, hi in zip(ladder, ladder[1:]):
                assert hc.hc_level_up(pre, run, lo) == hi, (pre, run, lo)
                assert hc.hc_level_down(pre, run, hi) == lo, (pre, run, hi)
            assert hc.hc_level_up(pre, run, D) == D and hc.hc_level_down(pre, run, S) == S
            # a level that does not exist at this geometry is never proposed
            for m in (S, P, R, D):
                assert hc.hc_level_up(pre, run, m) in ladder and hc.hc_level_down(pre, run, m) in ladder


def test_probes_that_cannot_succeed_are_ruled_out(hc):
    """rt_core.h: probe_ruled_out -- the sparse level while a stream has more cells over the absolute threshold than
    its lists hold; the chunk-bit level from the exact one while more than 3/4 of a stream's cells pass that threshold;
    nothing without a count, nothing for other moves."""
    D, S, P, R = _native.RT_MODE_DENSE, _native.RT_MODE_SPARSE, _native.RT_MODE_PREFILTER, _native.RT_MODE_RUNFILTER
    hc.hc_probe_ruled_out.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong]
    lists, cells = 16 * 1024, 1171 * 256
    assert hc.hc_probe_ruled_out(S, R, 1, lists + 1, lists, cells) == 1
    assert hc.hc_probe_ruled_out(S, P, 1, lists + 1, lists, cells) == 1
    assert hc.hc_probe_ruled_out(S, R, 1, lists, lists, cells) == 0       # exactly full still fits
    assert hc.hc_probe_ruled_out(S, R, 0, 10 * lists, lists, cells) == 0  # no count yet
    assert hc.hc_probe_ruled_out(S, S, 1, 10 * lists, lists, cells) == 0  # not a probe
    assert hc.hc_probe_ruled_out(P, R, 1, 3 * cells // 4 + 1, lists, cells) == 1
    assert hc.hc_probe_ruled_out(P, R, 1, 3 * cells // 4, lists, cells) == 0
    assert hc.hc_probe_ruled_out(P, R, 1, cells, lists, 0) == 0           # an empty call
    assert hc.hc_probe_ruled_out(R, D, 1, cells, lists, cells) == 0       # the dense level keeps no count: always probes
    assert hc.hc_probe_ruled_out(P, D, 1, cells, lists, cells) == 0


def test_quiet_level_samples_cover_at_least_32_segments(hc):
    """rt_core.h: minsum_group -- chunks per sample of the exact pre-filter's quiet-level estimate: the smallest power of
    two that makes 32 segments, at most the chunks of a workgroup's item."""
    hc.hc_minsum_group.argtypes = [C.c_int, C.c_int]
    assert [hc.hc_minsum_group(L, 16) for L in (4, 8, 16, 32, 71)] == [8, 4, 2, 1, 1]
    assert [hc.hc_minsum_group(L, 4) for L in (4, 8, 24, 37)] == [4, 4, 2, 1]
    assert hc.hc_minsum_group(4, 1) == 1


def test_quiet_level_margin_normalises_long_samples(hc):
    """rt_core.h: minsum_margin -- 1 up to 32 segments per sample, then the ratio that puts the expected minimum of longer
    samples (chunks of 37 .. 71 segments at nperseg >= 1024) where that of 32-segment samples lies."""
    hc.hc_minsum_margin.argtypes = [C.c_int]
    hc.hc_minsum_margin.restype = C.c_double
    assert hc.hc_minsum_margin(4) == 1.0 and hc.hc_minsum_margin(32) == 1.0
    for m, want in ((37, 0.963), (71, 0.847), (128, 0.785)):
        assert abs(hc.hc_minsum_margin(m) - want) < 2e-3, (m, hc.hc_minsum_margin(m))


def test_sampled_absolute_threshold_count_covers_every_chunk_length():
    """stft_scan MODE 6 counts the cells over the absolute threshold from one step in P (rt_core.h: abs_sample_period /
    abs_sampled) where nothing else needs the bits.  ADVICE round 5: a fixed period of eight keyed on the step number
    never sampled chunks of 4 .. 7 segments and kept one phase for every chunk when L is a multiple of eight."""
    import ctypes as C

    from pyradiotracking_amd import build

    lib = C.CDLL(build.build_hostcheck())
    for L, want in ((1, 1), (2, 2), (4, 4), (5, 4), (7, 4), (8, 8), (25, 8), (32, 8), (71, 8)):
        assert lib.hc_abs_sample_period(L) == want
    for L in (4, 5, 7, 8, 25, 32):
        P = lib.hc_abs_sample_period(L)
        weights = [lib.hc_abs_sampled_weight(L, ph) for ph in range(16)]
        # every chunk is sampled, and a sampled chunk stands for about its length (exactly, where P divides L)
        assert all(w >= P for w in weights), (L, weights)
        assert all(L - P < w < L + P for w in weights), (L, weights)
        if L % P == 0:
            assert set(weights) == {L}
    # the phase turns: the sampled segments of consecutive items / waves differ (L = 4: each of the four segments in turn;
    # L = 32: the residue mod 8 of the sampled segments walks through all eight values)
    assert sorted(lib.hc_abs_sampled_segment(4, ph, 0) for ph in range(4)) == [0, 1, 2, 3]
    assert sorted(lib.hc_abs_sampled_segment(32, ph, 0) % 8 for ph in range(8)) == list(range(8))
