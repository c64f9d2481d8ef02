"""The three selection rules of the scan kernel (rt_kernels.h: stft_scan), restated in NumPy and checked against the
plain definitions on random threshold maps -- the arguments DESIGN.md gives for them, as executable properties (CPU):

* candidate cells: a cell is emitted iff it passes the absolute threshold or directly precedes one that does; with the
  step below the chunk instead of a halo segment every such cell is emitted by exactly one chunk;
* sparse look-back tail: a tail cell is written iff the later cells of its chunk all pass the threshold -- every cell
  the reference's downward walk (analyze.py:382-398) can reach from the next buffer is among them;
* run-length pre-filter: chunks with an all-hot bit in themselves or a neighbour (chunk 0: the cells of the run through
  t = 0) contain every cell of every run that can pass the duration gate or continue a run of the previous buffer.

* exact run-length pre-filter, the planner (rt_core.h: RunPlanner / plan_tile_column, the arithmetic of the plan_runs kernel,
  compiled for the host): need = the cells of threshold runs of at least r cells and of the run through t = 0, plus the cell
  before each -- against that definition written out per bit, for every tiling and every counter width.

`hot[t]` below is "cell t of one bin passes the absolute threshold" (what the kernel keeps as bits per lane).
"""
import ctypes as C

import numpy as np
import pytest

L = 32  # segments per chunk


def _random_hot(rng, n_t):
    """a row of threshold bits with runs of every length, runs across chunk boundaries and at both ends"""
    p = rng.choice([0.02, 0.3, 0.6, 0.9, 0.98])
    hot = rng.random(n_t) < p
    for _ in range(int(rng.integers(0, 6))):  # planted long runs
        a = int(rng.integers(0, n_t))
        hot[a:a + int(rng.integers(1, 4 * L))] = True
    for _ in range(int(rng.integers(0, 3))):  # single cold cells at chunk edges
        c = int(rng.integers(0, max(1, n_t // L))) * L
        hot[min(n_t - 1, c + int(rng.integers(-1, 2)))] = False
    if rng.random() < 0.3:
        hot[-int(rng.integers(1, 3 * L)):] = True
    if rng.random() < 0.3:
        hot[:int(rng.integers(1, 3 * L))] = True
    return hot


def _chunks(n_t):
    return [(c0, min(c0 + L, n_t)) for c0 in range(0, n_t, L)]


@pytest.mark.parametrize("seed", range(40))
def test_every_candidate_cell_is_emitted_by_exactly_one_chunk(seed):
    rng = np.random.default_rng([1, seed])
    n_t = int(rng.integers(1, 8 * L))
    hot = _random_hot(rng, n_t)
    want = {t for t in range(n_t) if hot[t] or (t + 1 < n_t and hot[t + 1])}
    emitted = []
    for c0, c1 in _chunks(n_t):
        next_hot = False  # the chunk's first step knows nothing about the segment above it
        for t in range(c1 - 1, c0 - 1, -1):  # steps 1 .. L, descending time
            if hot[t] or next_hot:
                emitted.append(t)
            next_hot = hot[t]
        if c0 > 0 and hot[c0]:  # step L + 1 on the segment below: cells that precede a hot one and are not hot themselves
            if not hot[c0 - 1]:
                emitted.append(c0 - 1)
    assert sorted(emitted) == sorted(want)  # no cell twice, none missing


def _walk_reaches(hot_prev, k_cols):
    """tail cells (distance d = 1 .. from the end of the previous buffer) the downward walk can read: it reads d while
    every cell nearer to the end passed (cell_above implies hot), stops ON the first that does not (T11)"""
    reached = []
    n = len(hot_prev)
    for d in range(1, min(n, k_cols) + 1):
        reached.append(d)
        if not hot_prev[n - d]:
            break
    return reached


@pytest.mark.parametrize("seed", range(40))
def test_sparse_tail_holds_every_cell_a_walk_can_reach(seed):
    rng = np.random.default_rng([2, seed])
    n_t = int(rng.integers(1, 8 * L))
    hot = _random_hot(rng, n_t)
    k_cols = int(rng.integers(1, 2 * n_t + 2))
    written = set()
    for c0, c1 in _chunks(n_t):
        allhot = True  # of the chunk's later segments, all ones at its last one
        for t in range(c1 - 1, c0 - 1, -1):
            if allhot and t >= n_t - k_cols:
                written.add(t)
            allhot = allhot and hot[t]
    reached = {n_t - d for d in _walk_reaches(hot, k_cols)}
    assert reached <= written
    # ... and it is sparse where the input is: a column per chunk plus the runs that touch a chunk's end
    if not hot.any():
        assert len(written) == len([c for c in _chunks(n_t) if c[1] - 1 >= n_t - k_cols])


@pytest.mark.parametrize("seed", range(60))
def test_prefilter_chunks_contain_every_run_that_can_matter(seed):
    rng = np.random.default_rng([3, seed])
    n_t = int(rng.integers(2 * L, 10 * L))
    hot = _random_hot(rng, n_t)
    r_min = int(rng.integers(2 * L - 1, 4 * L))  # cells a run needs to pass the duration gate (the level needs 2 L - 1 <= r_min)
    chunks = _chunks(n_t)
    full = [bool(hot[c0:c1].all()) and (c1 - c0 == L) for c0, c1 in chunks]  # a partial last chunk never counts as all hot
    # pass A also keeps chunk 0 cell by cell; plan_pass_b and-s it up from t = 0: the cells of the run through t = 0
    prefix = np.logical_and.accumulate(hot[:L])
    emitted = set()
    for c, (c0, c1) in enumerate(chunks):
        need = full[c] or (c > 0 and full[c - 1]) or (c + 1 < len(chunks) and full[c + 1])
        next_hot = False
        for t in range(c1 - 1, c0 - 1, -1):
            need_t = need or (c == 0 and bool(prefix[t]))
            if need_t and (hot[t] or next_hot):
                emitted.add(t)
            next_hot = hot[t]
        if c0 > 0 and hot[c0] and need and not hot[c0 - 1]:
            emitted.add(c0 - 1)
    # maximal runs
    runs, b = [], None
    for t in range(n_t + 1):
        if t < n_t and hot[t]:
            b = t if b is None else b
        elif b is not None:
            runs.append((b, t))
            b = None
    for b, e in runs:
        if e == n_t:
            continue  # laps into the next buffer: skipped by the reference (analyze.py:415)
        if e - b >= r_min or b == 0:  # long enough, or it may continue a run of the previous buffer
            cells = set(range(b, e)) | ({b - 1} if b > 0 else set())  # + the cell the walk stops on (T11)
            assert cells <= emitted, (b, e, sorted(cells - emitted)[:5])


# ---------------------------------------------------------------------------
# the planner of the exact run-length pre-filter: bit-sliced counters against the definition
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hc():
    from pyradiotracking_amd import build
    lib = C.CDLL(build.build_hostcheck())
    lib.hc_plan_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.hc_plan_tile_rows.argtypes = [C.c_int, C.c_int, C.c_int]
    return lib


def _need_by_definition(hot_bits, r):
    """hot_bits [T][cells] bool -> need [T][cells]: C[t] = t lies in a run of >= r set cells (rows before the buffer count as
    set: the run through t = 0 may continue a plateau of the previous buffer, any length keeps it); need[t] = C[t] | C[t + 1]"""
    T, n = hot_bits.shape
    ext = np.concatenate([np.ones((r - 1, n), bool), hot_bits])
    c_ext = np.zeros_like(ext)
    for j in range(n):
        col = ext[:, j]
        edges = np.flatnonzero(np.diff(np.concatenate([[0], col.view(np.int8), [0]])))
        for b, e in zip(edges[0::2], edges[1::2]):
            if e - b >= r:
                c_ext[b:e, j] = True
    c = c_ext[r - 1:]
    need = c.copy()
    need[:-1] |= c[1:]
    return need


def _bits_to_words(bits, w):
    T = bits.shape[0]
    return np.packbits(bits.reshape(T, w, 64), axis=2, bitorder="little").view(np.uint64).reshape(T, w)


@pytest.mark.parametrize("seed", range(24))
def test_planner_counters_equal_the_run_length_definition(hc, seed):
    rng = np.random.default_rng([77, seed])
    w = int(rng.choice([4, 8, 16, 64]))                       # 64-bit words per row: nperseg 256 .. 4096
    T = int(rng.choice([1, 2, 9, 37, 150, 400, 1171]))
    r = int(rng.choice([1, 2, 9, 15, 16, 17, 63, 255, 256, 257, 300, 1200]))  # 4 / 8 / 16 planes, and r beyond the buffer
    tile = int(rng.choice([0, 0, 1, 7, 64, 100]))             # 0: the kernel's own tiling (rt_core.h: plan_tile_rows)
    bits = np.zeros((T, w * 64), bool)
    for j in range(w * 64):
        bits[:, j] = _random_hot(rng, T) if j % 7 else (rng.random(T) < rng.choice([0.0, 0.5, 1.0]))
    if rng.random() < 0.5:
        bits[:, : w * 8] = True                               # bins set throughout: runs longer than any counter
    hot = np.ascontiguousarray(_bits_to_words(bits, w))
    need = np.zeros_like(hot)
    planes = hc.hc_plan_runs(hot.ctypes.data, need.ctypes.data, T, w, r, tile)
    r_eff = min(r, T + 1)
    assert planes == (4 if r_eff <= 16 else 8 if r_eff <= 256 else 16)
    want = _bits_to_words(_need_by_definition(bits, r_eff), w)
    assert np.array_equal(need, want), (seed, w, T, r, tile)


def test_planner_tiles_fit_the_kernels_flags(hc):
    """a wave's rows (64 / w tiles side by side) stay within the byte flags it keeps in LDS, whatever the call"""
    for lg in (16, 32, 64, 128, 256):
        for n_seg in (1, 2, 100, 1171, 8000, 65535, 1 << 20):
            for r in (1, 9, 63, 127, 1000, 65536):
                rows = hc.hc_plan_tile_rows(n_seg, lg, min(r, n_seg + 1))
                tpw = 64 // (lg // 4)
                assert rows >= 1 and tpw * rows <= 16384 + tpw, (lg, n_seg, r, rows)
