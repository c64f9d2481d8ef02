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
