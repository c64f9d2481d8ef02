"""CPU restatement of the reference's cross-SDR matcher.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tests/perf/bench_match.py's CPU
baseline, never by the product (pyradiotracking_amd.match calls the C-ABI of
include/rt_match.h).

What it restates
  * SignalMatcher.add          radiotracking/match.py:54-82
  * MatchingSignal properties  radiotracking/__init__.py:293-335 (duration = max,
                               ts = min, frequency = statistics.median, per-device avgs)
  * MatchingSignal.has_member  radiotracking/__init__.py:337-387
  * MatchingSignal.add_member  radiotracking/__init__.py:389-406

It works on datetime / timedelta objects like the reference, so CPython itself
supplies the microsecond rounding (timedelta(seconds=..), timedelta / 2) that the
native code has to reproduce with integers.

Pinned: tests/golden/match_cases.npz holds inputs and the reference's own outputs
(tests/golden/make_golden_match.py imports radiotracking.match in the build
container); tests/test_match.py checks this module against them, and against the
imported reference directly when /root/reference is present.
"""
from __future__ import annotations

import datetime
import statistics
from typing import Dict, List, NamedTuple, Optional, Sequence


class MatchInput(NamedTuple):
    """The fields of a Signal the matcher reads."""

    device: str
    ts: datetime.datetime
    frequency: float
    duration: datetime.timedelta
    avg: float


class MatchGroup:
    """One open group: at most one member per device, insertion ordered."""

    def __init__(self, devices: Sequence[str]):
        self.devices = list(devices)
        self.members: Dict[str, MatchInput] = {}

    # __init__.py:293-335
    @property
    def duration(self) -> datetime.timedelta:
        return max(m.duration for m in self.members.values())

    @property
    def ts(self) -> datetime.datetime:
        return min(m.ts for m in self.members.values())

    @property
    def frequency(self) -> float:
        return statistics.median(m.frequency for m in self.members.values())

    @property
    def avgs(self) -> List[Optional[float]]:
        return [self.members[d].avg if d in self.members else None for d in self.devices]

    def accepts(self, sig: MatchInput, time_diff: datetime.timedelta, bandwidth: float,
                duration_diff: Optional[datetime.timedelta]) -> bool:
        """has_member (__init__.py:337-387): four interval tests, two more with a duration tolerance."""
        centre = self.frequency
        if sig.frequency - bandwidth / 2 > centre or sig.frequency + bandwidth / 2 < centre:
            return False
        first, span = self.ts, self.duration
        if sig.ts - time_diff > first + span:
            return False
        if (sig.ts + sig.duration) + time_diff < first:
            return False
        if duration_diff:
            if sig.duration - duration_diff / 2 > span or sig.duration + duration_diff / 2 < span:
                return False
        return True

    def take(self, sig: MatchInput) -> None:
        """add_member (__init__.py:389-406): a second signal of a device only replaces a quieter one."""
        have = self.members.get(sig.device)
        if have is None or have.avg < sig.avg:
            self.members[sig.device] = sig

    def snapshot(self):
        return (self.ts, self.frequency, self.duration, self.avgs, len(self.members))


class OracleMatcher:
    """match.py:32-82 on MatchInput tuples; ``add`` returns the groups consumed by that call."""

    def __init__(self, device: Sequence[str], matching_timeout_s: float, matching_time_diff_s: float,
                 matching_bandwidth_hz: float, matching_duration_diff_ms: Optional[float] = None):
        self.devices = list(device)
        self.timeout = datetime.timedelta(seconds=matching_timeout_s)
        self.time_diff = datetime.timedelta(seconds=matching_time_diff_s)
        self.bandwidth = float(matching_bandwidth_hz)
        self.duration_diff = (datetime.timedelta(milliseconds=matching_duration_diff_ms)
                              if matching_duration_diff_ms else None)
        self.open: List[MatchGroup] = []

    def add(self, sig: MatchInput) -> List[tuple]:
        out = []
        horizon = sig.ts - self.timeout
        for grp in list(self.open):
            if grp.ts < horizon:  # timed out: consumed, the walk goes on
                out.append(grp.snapshot())
                self.open.remove(grp)
                continue
            if grp.accepts(sig, self.time_diff, self.bandwidth, self.duration_diff):
                grp.take(sig)  # first match wins; later groups are not even checked for time-out
                return out
        grp = MatchGroup(self.devices)
        grp.take(sig)
        self.open.append(grp)
        return out

    def pending(self) -> List[tuple]:
        return [g.snapshot() for g in self.open]
