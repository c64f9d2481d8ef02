"""CPU-baseline leg of bench.py: the oracle (a port of the reference's
SciPy/NumPy path) timed on the GPU node's host cores, one worker process per
core, each analysing whole streams -- the reference's own parallelism model
(one process per SDR pinned to a core, radiotracking/__main__.py:118-128).

For the streams named in ``parity`` the worker also returns what the oracle
makes of the SAME buffer arriving a second time (look-back into the first
pass's spectrogram, analyze.py:383-388): bench.py analyses one resident
buffer over and over, so that is the state its records come from.  Only the
first pass is timed.

Kept import-light on purpose (no torch): workers are spawned."""
import datetime
import os
import time

import numpy as np

_TS0 = datetime.datetime(2024, 1, 1)
_iq = None


def _init(path):
    global _iq
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    _iq = np.load(path, mmap_mode="r")


def signal_rows(every, kept):
    """(fi, start, end, kept?, max, avg, std, noise, snr) per extracted signal, in the oracle's order"""
    kept_ids = {id(s) for s in kept}
    return [(s.fi, s.start, s.end, id(s) in kept_ids, s.max, s.avg, s.std, s.noise, s.snr) for s in every]


def _analyze(args):
    idx, kwargs, steady = args
    from oracle import analyze_oracle as oracle

    buf = np.array(_iq[idx])  # private copy, as a callback would get
    oa = oracle.OracleAnalyzer(device=str(idx), **kwargs)
    t0 = time.perf_counter()
    every, kept = oa.process(buf, _TS0)
    dt = time.perf_counter() - t0
    if steady:
        every, kept = oa.process(buf, _TS0)  # the same buffer again, with the first pass as `_spectrogram_last`
    return idx, dt, signal_rows(every, kept)


def run(iq_path: str, n_streams: int, kwargs: dict, workers: int, parity=()):
    """-> dict(wall_s, per_stream_s, results{idx: rows}).  The pool is started
    and warmed before the clock starts; ``parity`` = row indices analysed a
    second time in steady state (untimed: they are mapped after the clock stops)."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    parity = set(int(i) for i in parity)
    with ctx.Pool(workers, initializer=_init, initargs=(iq_path,)) as pool:
        pool.map(_analyze, [(0, kwargs, False)] * workers)  # warm-up: imports, page-in
        t0 = time.perf_counter()
        res = pool.map(_analyze, [(i, kwargs, False) for i in range(n_streams)], chunksize=1)
        wall = time.perf_counter() - t0
        steady = pool.map(_analyze, [(i, kwargs, True) for i in sorted(parity)], chunksize=1) if parity else []
    results = {r[0]: r[2] for r in res}
    results.update({r[0]: r[2] for r in steady})
    return dict(wall_s=wall, per_stream_s=[r[1] for r in res], results=results)
