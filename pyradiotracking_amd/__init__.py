"""MI355X-native signal-analysis path for pyradiotracking.

This package holds exactly one hot path of the reference: the per-buffer
``SignalAnalyzer.process_samples`` callback (STFT power -> threshold -> run
extraction with look-back -> shadow filter -> ``Signal`` records), rebuilt as
hand-written gfx950 HIP kernels behind a C-ABI (``include/rt_analyze.h``).

This module is the message model the path emits (reference
``radiotracking/__init__.py:13-22`` for the dB helpers and ``:110-202`` for
``Signal``).  Downstream consumers of the reference (matcher, CSV/MQTT sinks)
only look at the attributes / ``header`` / ``as_list`` / ``as_dict`` defined
here, so these are kept field-for-field compatible.  ``MatchedSignal`` /
``MatchingSignal`` (reference ``:205-406``) are the result types of the
cross-SDR matcher (``pyradiotracking_amd.match``).
"""
import datetime as _dt
import enum as _enum
import statistics as _statistics
from typing import Any, Dict, List, Optional, Union

import numpy as np

__version__ = "0.1.0"

__all__ = ["dB", "from_dB", "Signal", "StateMessage", "MatchedSignal", "MatchingSignal", "__version__"]


def dB(val):
    """Power ratio -> decibel.  dtype-preserving (float32 in, float32 out),
    exactly like the reference helper (radiotracking/__init__.py:13-17)."""
    return 10 * np.log10(val)


def from_dB(dB):  # noqa: N803 - argument name kept for keyword compatibility
    """Decibel -> power ratio (radiotracking/__init__.py:20-22)."""
    return 10 ** (dB / 10)


_SIGNAL_HEADER = [
    "Device",
    "Time",
    "Frequency",
    "Duration",
    "max (dBW)",
    "avg (dBW)",
    "std (dB)",
    "noise (dBW)",
    "snr (dB)",
]


class Signal:
    """One detected signal on one device/stream.

    Field-compatible with the reference record
    (radiotracking/__init__.py:110-202): ``device, ts, frequency, duration,
    max, avg, std, noise, snr``; ``ts`` accepts a datetime or an ISO string,
    ``duration`` a timedelta or seconds, the five power figures anything
    ``float()`` accepts.
    """

    header: List[str] = _SIGNAL_HEADER

    __slots__ = ("device", "ts", "frequency", "duration", "max", "avg", "std", "noise", "snr")

    def __init__(
        self,
        device: str,
        ts: Union[_dt.datetime, str],
        frequency: Union[float, str],
        duration: Union[_dt.timedelta, float, str],
        max_dBW: Union[float, str],  # noqa: N803
        avg_dBW: Union[float, str],  # noqa: N803
        std_dB: Union[float, str],  # noqa: N803
        noise_dBW: Union[float, str],  # noqa: N803
        snr_dB: Union[float, str],  # noqa: N803
    ):
        self.device = device
        self.ts = ts if isinstance(ts, _dt.datetime) else _dt.datetime.fromisoformat(ts)
        self.frequency = float(frequency)
        if isinstance(duration, _dt.timedelta):
            self.duration = duration
        else:
            self.duration = _dt.timedelta(seconds=float(duration))
        self.max = float(max_dBW)
        self.avg = float(avg_dBW)
        self.std = float(std_dB)
        self.noise = float(noise_dBW)
        self.snr = float(snr_dB)

    @property
    def as_list(self) -> List[Any]:
        return [getattr(self, name) for name in self.__slots__]

    @property
    def as_dict(self) -> Dict[str, Any]:
        return dict(zip(self.header, self.as_list))

    def __repr__(self) -> str:
        body = ", ".join(str(v) for v in self.as_list)
        return f"Signal({body})"

    def __str__(self) -> str:
        mhz = self.frequency / 1000 / 1000
        ms = self.duration.total_seconds() * 1000
        return f"Signal<SDR {self.device}, {mhz:.3f} MHz, {ms:.2f} ms, {self.max:.1f} dBW>"

    def __eq__(self, other) -> bool:
        return isinstance(other, Signal) and self.as_list == other.as_list

    def __hash__(self):
        return hash(tuple(self.as_list))


class StateMessage:
    """Liveness message of one analyzer (radiotracking/__init__.py:61-93): the analysis callback puts a
    STARTED message on the queue with its first buffer, RUNNING ones every ``state_update_s`` seconds
    after that and STOPPED when it gives up (analyze.py:180-190, 210-213, 226-229)."""

    class State(_enum.Enum):
        STOPPED = 0
        RUNNING = 1
        STARTED = 2

    header: List[str] = ["Device", "Time", "State"]

    def __init__(self, device: str, ts: _dt.datetime, state):
        self.device = device
        self.ts = ts
        self.state = state if isinstance(state, StateMessage.State) else StateMessage.State(int(state))

    @property
    def as_list(self) -> List[Any]:
        return [self.device, self.ts, self.state.value]

    @property
    def as_dict(self) -> Dict[str, Any]:
        return dict(zip(self.header, self.as_list))

    def __repr__(self) -> str:
        return f"StateMessage({self.device}, {self.ts}, {self.state})"


class MatchedSignal:
    """A signal seen on several devices of one station: earliest timestamp, median frequency,
    longest duration and one average power per device (``None`` where a device saw nothing).

    Constructor, ``header``, ``as_list``, ``as_dict``, ``repr`` and ``str`` follow the reference
    record (radiotracking/__init__.py:205-276)."""

    def __init__(
        self,
        devices: List[str],
        ts: Union[_dt.datetime, str],
        frequency: Union[float, str],
        duration: Union[_dt.timedelta, float, str],
        *avgs: Optional[float],
    ):
        self.devices = devices
        self._ts = ts if isinstance(ts, _dt.datetime) else _dt.datetime.fromisoformat(ts)
        self._frequency = float(frequency)
        self._duration = duration if isinstance(duration, _dt.timedelta) else _dt.timedelta(seconds=float(duration))
        self._avg_list: List[Optional[float]] = []
        for a in avgs:
            try:
                self._avg_list.append(float(a))
            except TypeError:  # None marks a device without a detection
                self._avg_list.append(None)

    @property
    def ts(self) -> _dt.datetime:
        return self._ts

    @property
    def frequency(self) -> float:
        return self._frequency

    @property
    def duration(self) -> _dt.timedelta:
        return self._duration

    @property
    def _avgs(self) -> List[Optional[float]]:
        return self._avg_list

    @property
    def header(self) -> List[str]:
        return ["Time", "Frequency", "Duration", *self.devices]

    @property
    def as_list(self) -> List[Any]:
        return [self.ts, self.frequency, self.duration, *self._avgs]

    @property
    def as_dict(self) -> Dict[str, Any]:
        return dict(zip(self.header, self.as_list))

    def __repr__(self) -> str:
        powers = ", ".join(repr(a) for a in self._avgs)
        return f"MatchedSignal({self.devices}, {self.ts}, {self.frequency}, {self.duration}, {powers})"

    def __str__(self) -> str:
        powers = ", ".join(f"{a:.2f}" if a else "None" for a in self._avgs)
        mhz = self.frequency / 1000 / 1000
        ms = self.duration.total_seconds() * 1000
        return f"{type(self).__name__}<SDRs {self.devices}, {mhz:.3f} MHz, {ms:.2f} ms, dBWs: [{powers}]>"


class MatchingSignal(MatchedSignal):
    """A group of per-device signals that is still collecting members
    (radiotracking/__init__.py:279-406): at most one ``Signal`` per device, the louder one on a
    repeat; ``ts`` / ``frequency`` / ``duration`` are min / median / max over the members.

    Groups handed out by :class:`pyradiotracking_amd.match.SignalMatcher` are snapshots of the
    native matcher's state: they carry the aggregated values and no member ``Signal`` objects."""

    def __init__(self, devices: List[str]):
        self.devices = devices
        self._sigs: Dict[str, Signal] = {}
        self._snapshot = None  # (ts, frequency, duration, avgs) of a group exported by the native matcher

    @classmethod
    def from_aggregate(cls, devices, ts, frequency, duration, avgs) -> "MatchingSignal":
        grp = cls(devices)
        grp._snapshot = (ts, float(frequency), duration, list(avgs))
        return grp

    @property
    def ts(self) -> _dt.datetime:
        return self._snapshot[0] if self._snapshot else min(s.ts for s in self._sigs.values())

    @property
    def frequency(self) -> float:
        return self._snapshot[1] if self._snapshot else _statistics.median(s.frequency for s in self._sigs.values())

    @property
    def duration(self) -> _dt.timedelta:
        return self._snapshot[2] if self._snapshot else max(s.duration for s in self._sigs.values())

    @property
    def _avgs(self) -> List[Optional[float]]:
        if self._snapshot:
            return self._snapshot[3]
        return [self._sigs[d].avg if d in self._sigs else None for d in self.devices]

    def has_member(self, sig: Signal, time_diff: _dt.timedelta = _dt.timedelta(0), bandwidth: float = 0,
                   duration_diff: Optional[_dt.timedelta] = None) -> bool:
        """Does ``sig`` overlap this group in frequency (+- bandwidth / 2), in time (+- time_diff)
        and, if ``duration_diff`` is given, in duration (+- duration_diff / 2)?"""
        centre, first, span = self.frequency, self.ts, self.duration
        if sig.frequency - bandwidth / 2 > centre or sig.frequency + bandwidth / 2 < centre:
            return False
        if sig.ts - time_diff > first + span or (sig.ts + sig.duration) + time_diff < first:
            return False
        if duration_diff and (sig.duration - duration_diff / 2 > span or sig.duration + duration_diff / 2 < span):
            return False
        return True

    def add_member(self, sig: Signal) -> None:
        if self._snapshot:
            raise TypeError("this group is a snapshot of the native matcher; it takes no members")
        have = self._sigs.get(sig.device)
        if have is None or have.avg < sig.avg:
            self._sigs[sig.device] = sig
