"""CPU-baseline leg of bench.py: the oracle (a port of the reference's
SciPy/NumPy path) timed on the GPU node's host cores, one worker process per
core, each analysing whole streams -- the reference's own parallelism model
(one process per SDR pinned to a core, radiotracking/__main__.py:118-128).

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


def _analyze(args):
    idx, kwargs = args
    from oracle import analyze_oracle as oracle

    buf = np.array(_iq[idx])  # private copy, as a callback would get
    t0 = time.perf_counter()
    every, kept = oracle.OracleAnalyzer(device=str(idx), **kwargs).process(buf, _TS0)
    dt = time.perf_counter() - t0
    keys = [(s.fi, s.start, s.end, s in kept) for s in every]
    return idx, dt, keys


def run(iq_path: str, n_streams: int, kwargs: dict, workers: int):
    """-> dict(wall_s, per_stream_s, results{idx: keys}).  The pool is started
    and warmed before the clock starts."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Pool(workers, initializer=_init, initargs=(iq_path,)) as pool:
        pool.map(_analyze, [(0, kwargs)] * workers)  # warm-up: imports, page-in
        t0 = time.perf_counter()
        res = pool.map(_analyze, [(i, kwargs) for i in range(n_streams)], chunksize=1)
        wall = time.perf_counter() - t0
    return dict(wall_s=wall, per_stream_s=[r[1] for r in res], results={r[0]: r[2] for r in res})
