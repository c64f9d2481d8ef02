"""Live cross-check of the oracle against the imported reference.

Runs only where /root/reference exists (the build container); the GPU box
relies on the committed golden vectors instead."""
import datetime
import os

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(
    not os.path.isdir("/root/reference/radiotracking"), reason="reference tree not present"
)


@pytest.fixture(scope="module")
def ref():
    from tests.golden import make_golden  # injects the rtlsdr stub and imports the reference

    return make_golden


@pytest.mark.parametrize("seed", range(6))
def test_random_streams_match_reference(ref, seed):
    import scipy.signal

    from oracle import analyze_oracle as oracle
    from pyradiotracking_amd import synth
    from tests import golden_util as gu

    rng = np.random.default_rng(1000 + seed)
    nperseg = int(rng.choice([256, 256, 1024]))
    fs = int(rng.choice([300000, 1000000, 2048000]))
    wname = str(rng.choice(["hamming", "hann"]))
    w = scipy.signal.get_window(wname, nperseg)
    blen = int(rng.integers(120, 400)) * nperseg + int(rng.integers(0, nperseg))
    nbuf = 3
    pulses = synth.random_pulses(rng, nbuf * blen, fs, w, 10, dur_ms=(6, 45), peak_dbw=(-88, -60))
    iq = synth.make_stream(synth.StreamSpec(nbuf * blen, fs, pulses, dc=complex(1e-3, 5e-4) * (seed % 2)), seed)
    bufs = [iq[i * blen : (i + 1) * blen] for i in range(nbuf)]
    tss = [gu.TS0 + datetime.timedelta(seconds=i * blen / fs) for i in range(nbuf)]
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=wname)

    want = ref.run_reference_buffers(ref.make_reference(**kw), bufs, tss)
    oa = oracle.OracleAnalyzer(device="0", **kw)
    n = 0
    for buf, ts, exp in zip(bufs, tss, want):
        every, kept = oa.process(buf, ts)
        tab = gu.signals_table(every, ts.replace(tzinfo=datetime.timezone.utc))
        assert np.array_equal(tab, exp["table"], equal_nan=True)
        kept_ids = {id(s) for s in kept}
        assert [id(s) in kept_ids for s in every] == list(exp["kept"])
        assert np.array_equal(oa.spec_last, exp["spec"])
        n += len(every)
    assert n > 0


@pytest.mark.parametrize("floor_db", [-2.0, 2.0, 8.0])
def test_noise_floor_around_the_threshold_matches_reference(ref, floor_db):
    """The regime in which decisions sit on the noise itself: the reference's default geometry and thresholds (300 kS/s,
    nperseg 256, -90 dBW, 5 dB SNR, 8 - 40 ms) with the noise floor 2 dB under, 2 dB over and 8 dB over the absolute
    threshold -- over it the SNR test alone decides.  Oracle and reference must agree exactly (same float32 powers)."""
    import scipy.signal

    from oracle import analyze_oracle as oracle
    from pyradiotracking_amd import synth
    from tests import golden_util as gu

    fs, nperseg, thr_dbw = 300000, 256, -90.0
    rng = np.random.default_rng(int(100 + floor_db))
    w = scipy.signal.get_window("hamming", nperseg)
    blen = 330 * nperseg + 17
    nbuf = 2
    sigma = float(np.sqrt(10.0 ** ((thr_dbw + floor_db) / 10.0) * fs / 2.0))  # PSD per bin = 2 sigma^2 / fs
    pulses = synth.random_pulses(rng, nbuf * blen, fs, w, 8, dur_ms=(9, 30), peak_dbw=(thr_dbw + floor_db + 14, thr_dbw + floor_db + 30))
    pulses.append(synth.Pulse(blen - int(0.006 * fs), int(0.015 * fs), 0.11 * fs, synth.amp_for_peak_dbw(thr_dbw + floor_db + 24, w, fs)))  # across the boundary
    iq = synth.make_stream(synth.StreamSpec(nbuf * blen, fs, pulses, noise_sigma=sigma), 7)
    bufs = [iq[i * blen : (i + 1) * blen] for i in range(nbuf)]
    tss = [gu.TS0 + datetime.timedelta(seconds=i * blen / fs) for i in range(nbuf)]
    kw = dict(sample_rate=fs, fft_nperseg=nperseg)

    want = ref.run_reference_buffers(ref.make_reference(**kw), bufs, tss)
    oa = oracle.OracleAnalyzer(device="0", **kw)
    n = 0
    for buf, ts, exp in zip(bufs, tss, want):
        every, kept = oa.process(buf, ts)
        tab = gu.signals_table(every, ts.replace(tzinfo=datetime.timezone.utc))
        assert np.array_equal(tab, exp["table"], equal_nan=True)
        kept_ids = {id(s) for s in kept}
        assert [id(s) in kept_ids for s in every] == list(exp["kept"])
        n += len(every)
    assert n > 5, n
