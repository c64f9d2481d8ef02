"""The multi-GPU story of this path is 'shard streams, no collective'.  These
CPU tests cover the partition arithmetic and the host-side record gather with a
world_size-2 gloo group."""
import os
import socket

import numpy as np
import pytest

from pyradiotracking_amd import _native, shard


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("n", [1, 7, 8, 256, 32768, 1000])
def test_partition_covers_once(world, n):
    spans = [shard.stream_range(r, world, n) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and a <= b and c <= d
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_streams, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.stream_range(rank, world, n_streams)
    # fabricate what a rank's analyzer returns: local stream ids, (fi, start) order
    rng = np.random.default_rng(rank)
    rec = np.zeros(0, dtype=_native.RECORD_DTYPE)
    rows = []
    for s in range(hi - lo):
        for k in range(int(rng.integers(0, 4))):
            r = np.zeros(1, dtype=_native.RECORD_DTYPE)
            r["stream"], r["fi"], r["start"], r["end"] = s, 10 * k + rank, k, k + 5
            rows.append(r)
    if rows:
        rec = np.concatenate(rows)
    merged = shard.gather_records(shard.to_global(rec, rank, world, n_streams))
    q.put((rank, lo, hi, len(rec), merged))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_gloo_world2():
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, lo0, hi0, n0, m0), (r1, lo1, hi1, n1, m1) = got
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)
    assert m0.tobytes() == m1.tobytes() and len(m0) == n0 + n1
    assert np.all(np.diff(m0["stream"]) >= 0)
    assert set(m0["stream"][:n0]) <= set(range(0, 6)) and set(m0["stream"][n0:]) <= set(range(6, 11))
