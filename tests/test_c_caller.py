"""The C-ABI from C: include/*.h are valid C99 and C++11, and a C program that binds librt_analyze.so directly
(tests/c/abi_smoke.c: rt_create / rt_process_host / rt_fetch / rt_get_call_info / rt_destroy) gets the records the
Python binding gets, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from pyradiotracking_amd import _native, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(REPO, "include")
PKG = os.path.join(REPO, "pyradiotracking_amd")


@pytest.mark.parametrize("compiler,std,lang", [("gcc", "c99", "c"), ("g++", "c++11", "c++")])
@pytest.mark.parametrize("header", ["rt_analyze.h", "rt_match.h", "rt_format.h"])
def test_headers_are_plain_c_and_cxx(compiler, std, lang, header):
    r = subprocess.run([compiler, f"-std={std}", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + INC, "-x", lang, "-"],
                       input=f'#include "{header}"\n', capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + INC, os.path.join(REPO, "tests", "c", "abi_smoke.c"),
                        "-L" + PKG, "-lrt_analyze", "-lm", "-Wl,-rpath," + PKG, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _window_file(tmp_path, fs, nperseg):
    from pyradiotracking_amd.analyze import stft_constants

    win32, scale32 = stft_constants("hamming", nperseg, fs)
    path = tmp_path / "window.bin"
    np.concatenate([win32, np.array([scale32], np.float32)]).astype(np.float32).tofile(path)
    return str(path)


def _streams(fs, nperseg, blen, n_streams):
    from pyradiotracking_amd.analyze import window_coefficients

    w = window_coefficients("hamming", nperseg)
    rng = np.random.default_rng(17)
    return np.stack([synth.make_stream(synth.StreamSpec(blen, fs, synth.random_pulses(rng, blen, fs, w, 4, keep_clear_tail=1024)), 300 + s)
                     for s in range(n_streams)])


def test_c_program_links_and_fails_loudly_without_a_gpu(tmp_path):
    """(CPU box) the program builds against the header and the library; without a GPU rt_create says RT_E_NO_DEVICE"""
    if _native.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    exe = _build(tmp_path)
    iq = _streams(2048000, 256, 256 * 40, 1)
    path = tmp_path / "iq.bin"
    iq.tofile(path)
    r = subprocess.run([exe, str(path), "1", str(iq.shape[1]), "2048000", "256", _window_file(tmp_path, 2048000, 256)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no GPU" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_program_gets_the_records_of_the_python_binding(tmp_path):
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer

    if _native.device_count() < 1:
        pytest.fail("no GPU visible")
    fs, nperseg, blen, n_streams = 2048000, 256, 256 * 900, 3
    exe = _build(tmp_path)
    iq = _streams(fs, nperseg, blen, n_streams)
    path = tmp_path / "iq.bin"
    iq.tofile(path)
    r = subprocess.run([exe, str(path), str(n_streams), str(blen), str(fs), str(nperseg), _window_file(tmp_path, fs, nperseg)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [tuple(ln.split()) for ln in r.stdout.splitlines()]
    an = BatchSignalAnalyzer([str(i) for i in range(n_streams)], sdr_callback_length=blen, sample_rate=fs, fft_nperseg=nperseg)
    an.enqueue(iq)
    rec = an.fetch_records()
    want = [(str(int(x["stream"])), str(int(x["fi"])), str(int(x["start"])), str(int(x["end"])), str(int(x["shadowed"])),
             *("%08x" % np.float32(x[k]).view(np.uint32) for k in ("max_p", "mean_p", "std_db", "row_mean"))) for x in rec]
    assert len(want) >= n_streams and got == want
