"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/rt_analyze.h declares.  No compute calls (no GPU here); what must
happen without a GPU is a loud failure, not a fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pyradiotracking_amd import _native, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _native.load_library()


def _declared_symbols(header="rt_analyze.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from pyradiotracking_amd import match

    assert _declared_symbols() == sorted(_native.ABI_SYMBOLS)
    assert _declared_symbols("rt_match.h") == sorted(match.MATCH_SYMBOLS)
    from pyradiotracking_amd import consume

    assert _declared_symbols("rt_format.h") == sorted(consume.FORMAT_SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    raw = C.CDLL(_native.LIB_PATH)
    headers = [h for h in os.listdir(os.path.join(REPO, "include")) if h.endswith(".h")]
    assert sorted(headers) == ["rt_analyze.h", "rt_format.h", "rt_match.h"]
    for header in headers:
        for name in _declared_symbols(header):
            assert hasattr(raw, name), name
    assert lib.rt_abi_version() == 6


def test_record_layout_matches_header():
    from pyradiotracking_amd import match

    assert _native.RECORD_DTYPE.itemsize == 40
    assert C.sizeof(_native.RtCallInfo) == 48
    assert match.SIGNAL_DTYPE.itemsize == 40 and match.MATCHED_DTYPE.itemsize == 32
    assert C.sizeof(match.RtMatchConfig) == 40
    from pyradiotracking_amd import consume

    assert consume.SIGNAL_ROW_DTYPE.itemsize == 72 and consume.MATCHED_ROW_DTYPE.itemsize == 24


def test_code_object_targets_gfx950():
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def _have_gpu(lib):
    n = C.c_int(0)
    lib.rt_device_count(C.byref(n))
    return n.value > 0


def test_no_gpu_fails_loudly(lib):
    if _have_gpu(lib):
        pytest.skip("a GPU is present")
    w = np.hamming(256).astype(np.float32)
    with pytest.raises(_native.NativeError) as ei:
        _native.NativeAnalyzer(
            n_streams=1, nperseg=256, max_samples=4096, sample_rate=300000.0, window_f32=w, scale=1.0,
            threshold=1e-9, snr_threshold=3.0, calibration_db=0.0, min_duration_s=0.008, max_duration_s=0.04,
        )
    assert ei.value.code == _native.RT_E_NO_DEVICE
    from pyradiotracking_amd.analyze import SignalAnalyzer

    with pytest.raises(_native.NativeError):
        SignalAnalyzer("0")


def test_argument_validation_happens_before_device_use(lib):
    cfg = _native.RtConfig()
    h = C.c_void_p()
    assert lib.rt_create(C.byref(cfg), C.byref(h)) == _native.RT_E_INVALID
    w = np.ones(9000, dtype=np.float32)
    cfg.n_streams, cfg.nperseg, cfg.max_samples, cfg.sample_rate = 1, 9000, 90000, 1e6  # (not a power of two and over 8192)
    cfg.window = w.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.rt_create(C.byref(cfg), C.byref(h)) == _native.RT_E_UNSUPPORTED
    assert b"nperseg" in lib.rt_last_error(None)
