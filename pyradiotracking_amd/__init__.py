"""MI355X-native signal-analysis path for pyradiotracking.

This package holds exactly one hot path of the reference: the per-buffer
``SignalAnalyzer.process_samples`` callback (STFT power -> threshold -> run
extraction with look-back -> shadow filter -> ``Signal`` records), rebuilt as
hand-written gfx950 HIP kernels behind a C-ABI (``include/rt_analyze.h``).

This module is the message model the path emits (reference
``radiotracking/__init__.py:13-22`` for the dB helpers and ``:110-202`` for
``Signal``).  Downstream consumers of the reference (matcher, CSV/MQTT sinks)
only look at the attributes / ``header`` / ``as_list`` / ``as_dict`` defined
here, so these are kept field-for-field compatible.
"""
import datetime as _dt
from typing import Any, Dict, List, Union

import numpy as np

__version__ = "0.1.0"

__all__ = ["dB", "from_dB", "Signal", "__version__"]


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
