"""One-subset-per-GPU sharding of independent streams.

Streams never read each other's data on this path, so a node is used as N
independent analyzers: rank r owns a contiguous block of streams, keeps their
IQ and look-back state in its own HBM, and no collective touches the data
path.  The only cross-rank traffic is the host-side concatenation of the
per-rank record lists (control plane, ``torch.distributed`` object gather).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def stream_range(rank: int, world: int, n_streams: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def to_global(records: np.ndarray, rank: int, world: int, n_streams: int) -> np.ndarray:
    """Shift a rank's local stream indices to global stream ids."""
    lo, _ = stream_range(rank, world, n_streams)
    out = records.copy()
    out["stream"] += lo
    return out


def gather_records(records_global: np.ndarray, group=None) -> np.ndarray:
    """All ranks' records (already carrying global stream ids) in stream order,
    on every rank.  Host-side only."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return records_global
    parts: List[np.ndarray] = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, records_global, group=group)
    return np.concatenate(parts) if parts else records_global
