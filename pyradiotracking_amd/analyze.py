"""Host-side mirror of the reference's ``SignalAnalyzer`` for the analysis path.

Same constructor keywords, same method names, same ``Signal`` records on the
same kind of queue as ``radiotracking/analyze.py:20-452`` -- but the arithmetic
of ``process_samples`` (STFT power, plateau extraction with look-back, shadow
verdicts) runs in the gfx950 HIP kernels behind ``librt_analyze.so``.  This
module only derives parameters, stages buffers and turns the integer/float32
records the kernels return into ``Signal`` objects (float64 / datetime parts
are computed here with the reference's own expressions, so they are bit-exact
by construction).

Not mirrored (out of scope): opening an RTL-SDR, the
SIGALRM watchdog, the process itself (the per-stream life-cycle rules of the
reference's Runner are in :mod:`pyradiotracking_amd.runner`).

Two entry levels:

* :class:`SignalAnalyzer` -- one stream, drop-in for the reference class.
* :class:`BatchSignalAnalyzer` -- ``S`` independent streams per call, IQ
  resident in HBM (``[S, B]`` complex64); this is what ``bench.py`` drives.
"""
from __future__ import annotations

import collections.abc
import datetime
import logging
import time
from typing import List, Optional, Sequence, Union

import numpy as np
import pytz
import scipy.fft
import scipy.signal

from . import Signal, StateMessage, dB, from_dB
from . import _native

logger = logging.getLogger(__name__)


def window_coefficients(fft_window, nperseg: int) -> np.ndarray:
    """What ``scipy.signal.spectrogram`` makes of its ``window`` argument
    (scipy/signal/_spectral_py.py:2239-2263): names/tuples go through
    ``get_window`` (periodic), arrays are taken as coefficients."""
    if isinstance(fft_window, (str, tuple)):
        return scipy.signal.get_window(fft_window, nperseg)
    win = np.asarray(fft_window)
    if win.ndim != 1:
        raise ValueError("window must be 1-D")
    if win.shape[0] != nperseg:
        raise ValueError("value specified for nperseg is different from length of window")
    return win


def stft_constants(fft_window, nperseg: int, sample_rate):
    """float32 window and PSD scale exactly as SciPy forms them for complex64
    input (_spectral_py.py:2083-2087): the window is cast to complex64 and
    ``scale = 1/(fs * sum(win*win))`` is evaluated in that dtype."""
    win = window_coefficients(fft_window, nperseg).astype(np.complex64)
    scale = 1.0 / (sample_rate * (win * win).sum())
    return np.ascontiguousarray(win.real, dtype=np.float32), np.float32(scale.real)


class _Heartbeat:
    """Stand-in for the ``multiprocessing.Value("d")`` the reference's Runner hands to every analyzer."""

    def __init__(self):
        self.value = 0.0


class _RecordDecoder:
    """rt_record arrays -> Signal field columns (analyze.py:360, 420-449)."""

    def __init__(self, nperseg: int, sample_rate, center_freq, calibration_db):
        """``calibration_db``: one value, or one per stream (the reference has one analyzer and one
        calibration per SDR, __main__.py:140-141)."""
        self.nperseg = nperseg
        self.sample_rate = sample_rate
        self.center_freq = center_freq
        self.calibration_db = calibration_db
        self.freqs = scipy.fft.fftfreq(nperseg, 1 / sample_rate)  # _spectral_py.py:2113

    def times(self, k: np.ndarray) -> np.ndarray:
        # element k of  arange(N/2, B - N/2 + 1, N) / float(fs)   (_spectral_py.py:2136)
        return (self.nperseg / 2 + np.asarray(k, dtype=np.float64) * self.nperseg) / float(self.sample_rate)

    def decode(self, rec: np.ndarray, freqs: Optional[np.ndarray] = None, times: bool = True):
        """Signal field columns of ``rec``; ``freqs`` replaces the analyzer's own frequency axis (caller-supplied
        spectrograms may have any number of bins, ``extract_signals``).  ``times=False``: without start time and duration
        (``None`` in their places -- the native row builder derives them from the cell coordinates itself)."""
        t_start = duration_s = None
        if times:
            start = rec["start"].astype(np.int64)
            end = rec["end"].astype(np.int64)
            t_end = self.times(end)
            t_start = np.where(start < 0, -self.times(np.abs(start)), self.times(np.maximum(start, 0)))
            duration_s = t_end - t_start
        cal = self.calibration_db
        if np.ndim(cal):
            # float32 dB figure minus a Python float is a float32 subtraction (NEP 50): the same bits as
            # subtracting the calibration rounded to float32
            cal = np.asarray(cal, dtype=np.float32)[rec["stream"]]
        with np.errstate(divide="ignore", invalid="ignore"):
            max_dbw = dB(rec["max_p"]) - cal  # float32, analyze.py:442
            avg_dbw = dB(rec["mean_p"]) - cal  # :444
            noise_dbw = dB(rec["row_mean"])  # :446
            snr_db = dB(rec["mean_p"] / rec["row_mean"])  # :447
        frequency = (self.freqs if freqs is None else np.asarray(freqs))[rec["fi"]] + self.center_freq  # :360
        return t_start, duration_s, frequency, max_dbw, avg_dbw, rec["std_db"], noise_dbw, snr_db

    def signal_columns(self, rec: np.ndarray, device_names: Sequence[str], ts_starts: Sequence[datetime.datetime]):
        """The nine ``Signal`` fields of every record of ``rec`` as nine Python lists (analyze.py:420-450), the reference's
        expressions evaluated column-wise: start and duration are whole hops, so a call holds few distinct values of either and
        one ``timedelta`` serves all records that share it; a stream's start time is taken to UTC once where its UTC offset
        does not change over the buffer (else record by record, as the reference does)."""
        t_start, duration_s, frequency, max_dbw, avg_dbw, std_db, noise_dbw, snr_db = self.decode(rec)

        def deltas(x):
            uniq, inv = np.unique(x, return_inverse=True)
            objs = np.empty(len(uniq), dtype=object)
            objs[:] = [datetime.timedelta(seconds=v) for v in uniq.tolist()]  # :428, :434
            return objs, inv

        start_td, start_i = deltas(t_start)
        dur_td, dur_i = deltas(duration_s)
        streams = rec["stream"]
        utc = pytz.utc
        # (ts_start + delta).astimezone(utc) == ts_start.astimezone(utc) + delta unless the zone's offset changes in between:
        # checked per stream at both ends of what the call can hold
        lo, hi = start_td[0], start_td[-1]
        n_names = len(device_names)
        base = np.empty(n_names, dtype=object)
        exact = True
        for s_ in np.unique(streams).tolist():
            b0 = ts_starts[s_]
            bu = b0.astimezone(utc)
            base[s_] = bu
            exact = exact and (b0 + lo).astimezone(utc) == bu + lo and (b0 + hi).astimezone(utc) == bu + hi
        starts = start_td[start_i].tolist()
        if exact:
            ts = [b + d for b, d in zip(base[streams].tolist(), starts)]  # :434, :449
        else:
            ts = [(ts_starts[s_] + d).astimezone(utc) for s_, d in zip(streams.tolist(), starts)]
        names = np.empty(n_names, dtype=object)
        names[:] = list(device_names)
        cols = [np.asarray(c, dtype=np.float64).tolist() for c in (frequency, max_dbw, avg_dbw, std_db, noise_dbw, snr_db)]  # float(np.float32): exact
        return [names[streams].tolist(), ts, cols[0], dur_td[dur_i].tolist()] + cols[1:]

    def signals(self, rec: np.ndarray, device_names: Sequence[str], ts_starts: Sequence[datetime.datetime]) -> List[Signal]:
        """``Signal`` objects of ``rec``: the columns above, the objects filled slot by slot (what ``Signal.__init__`` would store
        for these types, without its conversions)."""
        n = len(rec)
        if n == 0:
            return []
        new = Signal.__new__
        out = [new(Signal) for _ in range(n)]
        for sig, dv, ts, fr, du, mx, av, sd, no, sn in zip(out, *self.signal_columns(rec, device_names, ts_starts)):
            sig.device = dv
            sig.ts = ts
            sig.frequency = fr
            sig.duration = du
            sig.max = mx
            sig.avg = av
            sig.std = sd
            sig.noise = no
            sig.snr = sn
        return out

    def signal_batch(self, rec: np.ndarray, device_names: Sequence[str], ts_starts: Sequence[datetime.datetime]) -> "SignalBatch":
        """The same signals as a lazy sequence: the field columns are built at once, a ``Signal`` object only when an element
        is asked for -- for consumers that look at some of the signals, or hand the columns on (CSV rows, the matcher)."""
        if len(rec) == 0:
            return SignalBatch([[] for _ in range(9)])
        return SignalBatch(self.signal_columns(rec, device_names, ts_starts))


class SignalBatch(collections.abc.Sequence):
    """A read-only sequence of ``Signal`` over nine field columns (``_RecordDecoder.signal_batch``)."""

    __slots__ = ("columns",)

    def __init__(self, columns):
        self.columns = columns

    def __len__(self):
        return len(self.columns[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return SignalBatch([c[i] for c in self.columns])
        sig = Signal.__new__(Signal)
        (sig.device, sig.ts, sig.frequency, sig.duration, sig.max, sig.avg, sig.std, sig.noise, sig.snr) = [c[i] for c in self.columns]
        return sig

    def __eq__(self, other):
        """equal to any sequence of the same ``Signal`` values, a plain list included (``analyze_buffer(...) == []``)"""
        if not isinstance(other, (collections.abc.Sequence, SignalBatch)) or isinstance(other, (str, bytes)):
            return NotImplemented
        return len(self) == len(other) and all(_same_signal(a, b) for a, b in zip(self, other))

    __hash__ = None

    def __repr__(self):
        return f"SignalBatch({list(self)!r})"


def _same_signal(a, b) -> bool:
    fields = ("device", "ts", "frequency", "duration", "max", "avg", "std", "noise", "snr")
    try:
        return all(getattr(a, f) == getattr(b, f) or (getattr(a, f) != getattr(a, f) and getattr(b, f) != getattr(b, f)) for f in fields)
    except AttributeError:
        return False


def default_lanes(fft_nperseg: int, n_streams: int) -> int:
    """Stream groups per GPU (``rt_config.lanes``) that measured best: three up to nperseg 512 while the launches of a group are
    small enough to have ends worth filling by another group's kernels (fewer than 16 384 streams on the GPU) -- config 2 +14 % with
    two lanes and another +1.5 % with three, the reference's defaults under a noise floor +3 ... 4 % over two; four lanes -16 % at
    config 2 (the HIP runtime's four hardware queues: with GPU_MAX_HW_QUEUES=8 four lanes equal three, six lose); one lane from nperseg 1024 on, where the scans are chip-filling grids of persistent workgroups (config 3: 676 k
    MSamples/s with one lane, 650 k with two).  ``profiles/r05_n_lanes_by_batch_size.txt``; ``bench.py`` uses the same rule."""
    return 3 if (fft_nperseg <= 512 and 3 <= n_streams < 16384) else 1


class BatchSignalAnalyzer:
    """``S`` independent analyzers sharing one configuration, one GPU.

    ``process_batch`` is the batched body of the reference callback
    (analyze.py:231-251, 268) for one buffer of every stream.  Per-stream
    carried state (the look-back tail replacing ``_spectrogram_last``) lives in
    HBM inside the native handle.
    """

    def __init__(
        self,
        devices: Sequence[str],
        calibration_db: Union[float, Sequence[float]] = 0.0,
        sample_rate: int = 300000,
        center_freq: int = 150150000,
        fft_nperseg: int = 256,
        fft_window="hamming",
        signal_min_duration_ms: float = 8,
        signal_max_duration_ms: float = 40,
        signal_threshold_dbw: float = -90.0,
        snr_threshold_db: float = 5.0,
        sdr_callback_length: Optional[int] = None,
        gpu: int = 0,
        mode: str = "auto",
        hot_capacity: int = 0,
        record_capacity: int = 0,
        segs_per_chunk: int = 0,
        timing: bool = False,
        hip_stream: Optional[int] = None,
        lanes: Union[int, str] = 1,
        subtract_first: bool = False,
        record_pool: int = 0,
        group_detect: Optional[bool] = None,
        **kwargs,
    ):
        """``mode`` (``rt_config.mode``): ``"auto"`` (default) analyses on the fused sparse path and, when an input's noise
        crosses the thresholds, climbs by itself -- chunk-bit pre-filter, exact SNR-aware pre-filter, dense spectrogram; the
        records are the same on every level.  ``"sparse"``, ``"prefilter"``, ``"runfilter"`` and
        ``"dense"`` pin one level (the first three refuse an input they cannot hold with ``RT_E_HOT_OVERFLOW``).

        ``record_pool`` (``rt_config.record_pool``): records the pinned result pool holds at first (0: up to 4 Mi); a
        buffer with more signals grows it -- the reference appends without limit (``analyze.py:449-450``); so does the
        per-stream ``record_capacity`` (where a stream's room STARTS, default 1024): nothing bounds a call but memory.

        ``subtract_first``: apply SciPy's ``detrend='constant'`` in SciPy's order (segment mean subtracted before
        the window) even for hamming / hann / boxcar windows, where the kernels by default subtract ``mean * FFT(window)``
        from the three bins it touches instead (equal within float32 round-off, fewer operations).

        ``group_detect`` (``RT_FLAG_GROUP_DETECT`` / ``RT_FLAG_NO_GROUP_DETECT``): sparse detection with one wave per stream or
        quarter of a stream's sixteen candidate lists instead of one per list -- the same records; ``None`` (default) lets the
        library decide (from 1 024 streams per handle on, where the lists are many and short).

        ``lanes`` > 1 (``rt_config.lanes``) splits the streams into that many contiguous groups, each
        analysed on its own HIP stream: the detection kernels and launch gaps of one group then overlap the
        scan of another (config 2: +14 % whole-path throughput with two lanes).  Streams are independent,
        so the records are the same; ``rt_fetch`` returns them in stream order.  Needs ``hip_stream=None``.
        ``lanes="auto"`` takes what the sweeps of round 5 found best (``default_lanes``: three up to nperseg 512 while the
        batch holds fewer than 16 384 streams, one otherwise and whenever a ``hip_stream`` is given).

        ``calibration_db`` may be a sequence with one value per stream: every SDR of the reference has its own
        analyzer and calibration (``__main__.py:140-141``), and with it its own absolute threshold
        (``analyze.py:115``); the kernels then take the thresholds per stream (``rt_set_stream_params``)."""
        self.devices = [str(d) for d in devices]
        per_stream_cal = None
        if np.ndim(calibration_db):
            per_stream_cal = [float(c) for c in calibration_db]
            if len(per_stream_cal) != len(self.devices):
                raise ValueError(f"calibration values {per_stream_cal} do not match devices {self.devices}")  # __main__.py:219
            calibration_db = per_stream_cal[0]
        self.calibration_db = calibration_db
        self.sample_rate = sample_rate
        self.center_freq = center_freq
        if sdr_callback_length is None:  # analyze.py:108-109
            sdr_callback_length = sample_rate
        self.sdr_callback_length = int(sdr_callback_length)
        self.fft_nperseg = fft_nperseg
        self.fft_window = fft_window
        self.signal_min_duration = signal_min_duration_ms / 1000  # :113
        self.signal_max_duration = signal_max_duration_ms / 1000  # :114
        self.signal_threshold = from_dB(signal_threshold_dbw + calibration_db)  # :115
        self.snr_threshold = from_dB(snr_threshold_db)  # :116

        win32, scale32 = stft_constants(fft_window, fft_nperseg, sample_rate)
        if lanes == "auto":
            lanes = 1 if hip_stream is not None else default_lanes(fft_nperseg, len(self.devices))
        if int(lanes) > 1 and hip_stream is not None:
            raise ValueError("lanes > 1 run on their own HIP streams: pass hip_stream=None")
        self._native = _native.NativeAnalyzer(
            n_streams=len(self.devices),
            nperseg=fft_nperseg,
            max_samples=self.sdr_callback_length,
            sample_rate=sample_rate,
            window_f32=win32,
            scale=float(scale32),
            # thresholds are compared against float32 data in float32
            threshold=float(np.float32(self.signal_threshold)),
            snr_threshold=float(np.float32(self.snr_threshold)),
            calibration_db=calibration_db,
            min_duration_s=self.signal_min_duration,
            max_duration_s=self.signal_max_duration,
            device=gpu,
            mode={"auto": _native.RT_MODE_AUTO, "dense": _native.RT_MODE_DENSE, "sparse": _native.RT_MODE_SPARSE, "prefilter": _native.RT_MODE_PREFILTER,
                  "runfilter": _native.RT_MODE_RUNFILTER}[mode],
            hot_capacity=hot_capacity,
            record_capacity=record_capacity,
            segs_per_chunk=segs_per_chunk,
            timing=timing,
            hip_stream=hip_stream,
            lanes=max(1, int(lanes)),
            subtract_first=bool(subtract_first),
            record_pool=int(record_pool),
            group_detect=group_detect,
        )
        if per_stream_cal is not None:
            self.calibration_db = per_stream_cal
            self.signal_threshold = [from_dB(signal_threshold_dbw + c) for c in per_stream_cal]  # :115, per SDR
            self._native.set_stream_params(
                np.array(self.signal_threshold, dtype=np.float64).astype(np.float32), np.array(per_stream_cal, dtype=np.float32)
            )
        self._decoder = _RecordDecoder(fft_nperseg, sample_rate, center_freq, self.calibration_db)
        self.decoder = self._decoder  # record -> field conversion, shared with pyradiotracking_amd.match
        self.gpu = gpu
        self._hip_stream = hip_stream

    # -- native plumbing ----------------------------------------------------
    @property
    def native(self) -> "_native.NativeAnalyzer":
        return self._native

    def _hold(self, tensor):
        # two calls may be in flight: keep the device buffers of both alive (torch's caching
        # allocator would otherwise hand the memory to someone else while the scan still reads it)
        held = getattr(self, "_held", [])
        self._held = (held + [tensor])[-2:]

    def reset(self):
        """``_spectrogram_last = None`` for every stream."""
        self._native.reset()

    def reset_stream(self, stream: int):
        """``_spectrogram_last = None`` for one stream: its SDR was restarted, i.e. the reference would have
        replaced its analyzer by a fresh one (``__main__.py:185-190``)."""
        self._native.reset_stream(stream)

    def close(self):
        """Destroy the native handle (waits for everything in flight), then let go of the device tensors the
        calls in flight may have been reading."""
        self._native.close()
        self._held = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call_info(self):
        """Figures of the last fetched call (with lanes: counts and times summed over the lanes' launches)."""
        return self._native.call_info()

    def _process_device(self, ptr: int, n_samples: int, stride: Optional[int], bytes_per_sample: int, u8: bool):
        if u8:
            self._native.process_device_u8(ptr, n_samples, stride)
        else:
            self._native.process_device(ptr, n_samples, stride)

    def enqueue(self, iq, n_samples: Optional[int] = None, stream_stride: Optional[int] = None):
        """Start analysing one buffer per stream (asynchronous).

        ``iq``: a CUDA/HIP torch tensor ``[S, B]`` complex64, a raw device
        pointer (int, with ``n_samples``), or a host ndarray ``[S, B]``.
        """
        if isinstance(iq, int):
            if n_samples is None:
                raise ValueError("n_samples is required with a raw device pointer")
            self._process_device(iq, n_samples, stream_stride, 8, False)
            return
        if isinstance(iq, np.ndarray) and iq.dtype == np.uint8:
            self.enqueue_bytes(iq)
            return
        if isinstance(iq, np.ndarray):
            self._native.process_host(iq)
            return
        # torch tensor
        if iq.dim() == 1:
            iq = iq[None, :]
        if not iq.is_cuda:
            self.enqueue(iq.numpy())
            return
        if str(iq.dtype) != "torch.complex64":
            raise TypeError("device IQ must be complex64")
        if iq.shape[0] != len(self.devices) or iq.stride(1) != 1:
            raise ValueError("device IQ must be [S, B] with unit sample stride")
        self._hold(iq)
        if self._hip_stream is None:
            # the handle launches on its own stream: whatever torch still has in flight on
            # the producing stream must have landed before the scan kernel reads the IQ
            import torch

            torch.cuda.current_stream(iq.device).synchronize()
        self._process_device(iq.data_ptr(), iq.shape[1], iq.stride(0) if iq.shape[0] > 1 else iq.shape[1], 8, False)

    def enqueue_bytes(self, raw, n_samples: Optional[int] = None, stream_stride: Optional[int] = None):
        """Same as :meth:`enqueue` for the RTL-SDR wire format: interleaved uint8 I, Q (what
        librtlsdr delivers; pyrtlsdr's ``packed_bytes_to_iq`` turns it into the complex buffer
        the reference's callback sees).  ``raw``: host ndarray / CUDA torch tensor of uint8 shaped
        ``[S, 2*B]`` (or ``[2*B]`` for one stream), or a raw device pointer with ``n_samples``.
        The byte -> float conversion is fused into the scan kernel's load."""
        if isinstance(raw, int):
            if n_samples is None:
                raise ValueError("n_samples is required with a raw device pointer")
            self._process_device(raw, n_samples, stream_stride, 2, True)
            return
        if isinstance(raw, np.ndarray):
            a = np.ascontiguousarray(raw, dtype=np.uint8)
            if a.ndim == 1:
                a = a[None, :]
            if a.shape[0] != len(self.devices) or a.shape[1] % 2:
                raise ValueError("expected uint8 [S, 2*B]")
            if a.shape[1] // 2 > self.sdr_callback_length:
                raise ValueError("buffer longer than sdr_callback_length")
            # staged by the library: one device buffer per call in flight, kept until the call is fetched
            self._native.process_host_u8(a)
            return
        if raw.dim() == 1:
            raw = raw[None, :]
        if str(raw.dtype) != "torch.uint8" or not raw.is_cuda or raw.stride(1) != 1 or raw.shape[1] % 2:
            raise TypeError("device bytes must be a CUDA uint8 tensor [S, 2*B] with unit stride")
        self._hold(raw)
        if self._hip_stream is None:
            import torch

            torch.cuda.current_stream(raw.device).synchronize()
        stride = (raw.stride(0) if raw.shape[0] > 1 else raw.shape[1]) // 2
        self._process_device(raw.data_ptr(), raw.shape[1] // 2, stride, 2, True)

    def fetch_records(self, allow_truncated: bool = False) -> np.ndarray:
        """Wait for the oldest enqueued call; structured array of ``rt_record`` ordered by stream.  The reference
        has no limit on signals per buffer, and neither has an analysed buffer here (a stream that outgrows
        ``record_capacity`` grows it; the call is analysed again inside the fetch).  Only ``extract_signals`` on a caller's
        spectrogram can be cut off: that raises unless ``allow_truncated`` (then ``native.last_truncated`` tells)."""
        return self._native.fetch(allow_truncated)

    def process_batch(self, iq, ts_starts: Union[datetime.datetime, Sequence[datetime.datetime]], filtered: bool = True, lazy: bool = True):
        """One buffer per stream -> per stream a sequence of ``Signal``.

        ``filtered=True`` returns what the reference puts on its queue
        (after ``filter_shadow_signals``); ``False`` returns the list
        ``extract_signals`` would have produced.

        ``lazy`` (default): every stream's sequence is a :class:`SignalBatch` -- the nine field columns of the whole call are built
        at once, a ``Signal`` object only when an element is asked for (it compares equal to the list of those objects; 2.6 x the
        rate of building every object up front, and a consumer that hands columns on -- CSV rows, the matcher -- never needs
        them).  ``lazy=False``: plain lists of ``Signal`` objects."""
        self.enqueue(iq)
        rec = self.fetch_records()
        if isinstance(ts_starts, datetime.datetime):
            ts_starts = [ts_starts] * len(self.devices)
        if filtered:
            rec = rec[rec["shadowed"] == 0]
        # (records come ordered by stream: a stream's signals are one slice)
        bounds = np.searchsorted(rec["stream"], np.arange(len(self.devices) + 1))
        if lazy:
            batch = self._decoder.signal_batch(rec, self.devices, ts_starts)
            return [batch[int(bounds[s]):int(bounds[s + 1])] for s in range(len(self.devices))]
        sigs = self._decoder.signals(rec, self.devices, ts_starts)
        return [sigs[int(bounds[s]):int(bounds[s + 1])] for s in range(len(self.devices))]


class SignalAnalyzer:
    """Drop-in for ``radiotracking.analyze.SignalAnalyzer`` on the analysis path.

    Constructor keywords are the reference's (analyze.py:62-83; unknown ones are
    swallowed like its ``**kwargs``) plus ``gpu`` (HIP device ordinal).  The
    object is *not* a ``multiprocessing.Process`` and opens no SDR: feed it
    buffers through :meth:`process_samples` exactly as ``read_samples_async``
    would (analyze.py:157).
    """

    def __init__(
        self,
        device: str,
        calibration_db: float = 0.0,
        sample_rate: int = 300000,
        center_freq: int = 150150000,
        gain: float = 49.6,
        fft_nperseg: int = 256,
        fft_window="hamming",
        signal_min_duration_ms: float = 8,
        signal_max_duration_ms: float = 40,
        signal_threshold_dbw: float = -90.0,
        snr_threshold_db: float = 5.0,
        verbose: int = 0,
        sdr_max_restart: int = 3,
        sdr_timeout_s: int = 2,
        state_update_s: int = 60,
        sdr_callback_length: Optional[int] = None,
        signal_queue=None,
        last_data_ts=None,
        gpu: int = 0,
        mode: str = "auto",
        **kwargs,
    ):
        self.device = device
        self.calibration_db = calibration_db
        try:
            self.device_index = int(device)  # analyze.py:89-91
        except ValueError:
            self.device_index = None  # serial-number lookup needs pyrtlsdr: out of scope
        self.sample_rate = sample_rate
        self.center_freq = center_freq
        try:
            self.gain = float(gain)
        except ValueError:
            self.gain = gain
        if sdr_callback_length is None:
            sdr_callback_length = sample_rate
        self.fft_nperseg = fft_nperseg
        self.fft_window = fft_window
        self.signal_min_duration = signal_min_duration_ms / 1000
        self.signal_max_duration = signal_max_duration_ms / 1000
        self.signal_threshold = from_dB(signal_threshold_dbw + calibration_db)
        self.snr_threshold = from_dB(snr_threshold_db)
        self.sdr_callback_length = sdr_callback_length
        self.verbose = verbose
        self.sdr_max_restart = sdr_max_restart
        self.sdr_timeout_s = sdr_timeout_s
        self.state_update_s = state_update_s
        self.signal_queue = signal_queue
        # the reference always gets a multiprocessing.Value("d") from its Runner (__main__.py:118); without one the
        # heartbeat lives in a private holder, so that the first buffer reports STARTED and the later ones RUNNING
        self.last_data_ts = last_data_ts if last_data_ts is not None else _Heartbeat()
        self.last_state: Optional[StateMessage] = None
        self.sdr = None  # the caller's SDR handle, if it wants cancel_read_async() on a fatal clock drift

        self._spectrogram_last = None  # only used by extract_signals() called directly
        self._ts = None
        self._batch = BatchSignalAnalyzer(
            [device],
            calibration_db=calibration_db,
            sample_rate=sample_rate,
            center_freq=center_freq,
            fft_nperseg=fft_nperseg,
            fft_window=fft_window,
            signal_min_duration_ms=signal_min_duration_ms,
            signal_max_duration_ms=signal_max_duration_ms,
            signal_threshold_dbw=signal_threshold_dbw,
            snr_threshold_db=snr_threshold_db,
            sdr_callback_length=sdr_callback_length,
            gpu=gpu,
            mode=mode,
            # capacities of the native handle (no counterpart in the reference, whose lists are unbounded)
            **{k: kwargs[k] for k in ("record_capacity", "record_pool", "hot_capacity", "group_detect") if k in kwargs},
        )
        self._decoder = self._batch._decoder

    # -- the callback (analyze.py:192-268) -----------------------------------
    def update_state(self, ts: datetime.datetime, state: "StateMessage.State") -> None:
        """Put a ``StateMessage`` on the queue unless the same state was reported less than
        ``state_update_s`` seconds ago (analyze.py:180-190)."""
        ts = ts.astimezone(pytz.utc)
        last = self.last_state
        if last and last.state == state and last.ts + datetime.timedelta(seconds=self.state_update_s) >= ts:
            return
        self.last_state = StateMessage(self.device, ts, state)
        if self.signal_queue is not None:
            self.signal_queue.put(self.last_state)

    def _clock(self, n_samples: int):
        """Book-keeping at the head of the callback (analyze.py:204-231): liveness, heartbeat value,
        running clock, drift check.  Returns ``ts_start`` of the buffer."""
        ts_recv = datetime.datetime.now()
        buffer_len_dt = datetime.timedelta(seconds=n_samples / self.sample_rate)  # :205
        if not self.last_data_ts.value:  # :210-213
            self.update_state(datetime.datetime.now(), StateMessage.State.STARTED)
        else:
            self.update_state(ts_recv, StateMessage.State.RUNNING)
        self.last_data_ts.value = datetime.datetime.timestamp(ts_recv)  # :214
        if not self._ts:  # :218-221
            self._ts = ts_recv
        else:
            self._ts += buffer_len_dt
        clock_drift = (ts_recv - self._ts).total_seconds()
        if clock_drift > 2 * buffer_len_dt.total_seconds():  # :226-229
            logger.warning(
                f"SDR {self.device} total clock drift ({clock_drift:.5f} s) is larger than two blocks, "
                "signal detection is degraded. Terminating..."
            )
            self.update_state(datetime.datetime.now(), StateMessage.State.STOPPED)
            if self.sdr is not None:
                self.sdr.cancel_read_async()
        return self._ts - buffer_len_dt  # :231

    def process_samples(self, buffer: np.ndarray, context=None):
        """Analyse one buffer; state messages and detected signals go to ``signal_queue`` in the
        reference's order.  complex128 buffers (what pyrtlsdr delivers) are analysed in complex64 --
        the GPU path is single precision.  The SIGALRM watchdog of the reference
        (analyze.py:208) belongs to the process that owns the SDR and is not re-armed here."""
        ts_start = self._clock(len(buffer))
        filtered = self.analyze_buffer(buffer, ts_start)
        [self.consume_signal(s) for s in filtered]  # :251
        return None

    def process_bytes(self, raw: np.ndarray, context=None):
        """The callback for ``RtlSdr.read_bytes_async``: ``raw`` is the interleaved uint8 I/Q
        buffer (2 bytes per sample).  Same bookkeeping and queue contract as
        :meth:`process_samples`; the conversion pyrtlsdr would do on the host happens in the
        scan kernel's load."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        n = raw.size // 2
        ts_start = self._clock(n)
        if n > self._batch.sdr_callback_length:
            raise ValueError("buffer longer than sdr_callback_length")
        self._batch.enqueue_bytes(raw.reshape(1, -1))
        rec = self._batch.fetch_records()
        rec = rec[rec["shadowed"] == 0]
        [self.consume_signal(s) for s in self._decoder.signals(rec, [self.device], [ts_start])]
        return None

    def analyze_buffer(self, buffer: np.ndarray, ts_start: datetime.datetime, filtered: bool = True) -> List[Signal]:
        """analyze.py:234-248 and :268 with an explicit ``ts_start``."""
        bench_start = time.time()
        buf = np.ascontiguousarray(buffer, dtype=np.complex64).reshape(1, -1)
        if buf.shape[1] > self._batch.sdr_callback_length:
            raise ValueError("buffer longer than sdr_callback_length")
        out = self._batch.process_batch(buf, [ts_start], filtered=filtered, lazy=False)[0]  # (one stream: its Signal objects, as the reference returns them)
        logger.info(
            f"SDR {self.device} recv {len(buffer)}, {len(out)} signals, "
            f"compute: {(time.time() - bench_start) * 1000:.1f} ms"
        )
        return out

    def reset(self):
        self._batch.reset()
        self._spectrogram_last = None
        self._ts = None
        if isinstance(self.last_data_ts, _Heartbeat):
            self.last_data_ts.value = 0.0

    # -- the reference's public helpers ---------------------------------------
    def consume_signal(self, signal: Signal):
        """analyze.py:270-280."""
        logger.debug(f"SDR {self.device} received {signal}")
        if self.signal_queue is not None:
            self.signal_queue.put(signal)

    def extract_signals(self, freqs: np.ndarray, times: np.ndarray, spectrogram: np.ndarray, ts_start: datetime.datetime) -> List[Signal]:
        """analyze.py:330-452 on an explicit ``[F, T]`` power spectrogram (any
        ``F``), with ``self._spectrogram_last`` as the previous one.  Runs the
        dense detect kernel (``rt_extract``).  The time axis must be the one
        SciPy produces for ``fft_nperseg`` / ``sample_rate`` (hop = nperseg/fs);
        ``freqs`` is used as given."""
        spec = np.asarray(spectrogram)
        n_bins, n_seg = spec.shape
        if n_seg == 0:
            return []
        if n_seg >= 2:
            hop = self._decoder.times(np.array([1]))[0] - self._decoder.times(np.array([0]))[0]
            if not np.isclose(times[1] - times[0], hop, rtol=1e-9, atol=0.0):
                raise ValueError("times does not match fft_nperseg / sample_rate")
        nat = self._batch.native
        seg_major = np.ascontiguousarray(spec.T, dtype=np.float32)
        d_spec = _native.DeviceBuffer(self._batch.gpu, max(4, seg_major.nbytes))
        d_spec.upload(seg_major)
        d_last = None
        n_last = 0
        if self._spectrogram_last is not None:
            last = np.ascontiguousarray(np.asarray(self._spectrogram_last).T, dtype=np.float32)
            n_last = last.shape[0]
            d_last = _native.DeviceBuffer(self._batch.gpu, max(4, last.nbytes))
            d_last.upload(last)
        try:
            nat.extract_device(d_spec.ptr, n_seg, n_bins, d_last.ptr if d_last else None, n_last)
            rec = nat.fetch()
        finally:
            d_spec.free()
            if d_last:
                d_last.free()
        self._last_records = rec
        t_start, duration_s, frequency, max_dbw, avg_dbw, std_db, noise_dbw, snr_db = self._decoder.decode(rec, freqs)
        out = []
        for i in range(len(rec)):
            ts = ts_start + datetime.timedelta(seconds=float(t_start[i]))
            out.append(
                Signal(
                    self.device,
                    ts.astimezone(pytz.utc),
                    frequency[i],
                    datetime.timedelta(seconds=float(duration_s[i])),
                    max_dbw[i],
                    avg_dbw[i],
                    std_db[i],
                    noise_dbw[i],
                    snr_db[i],
                )
            )
        return out

    @staticmethod
    def is_shadow_of(sig: Signal, signals: List[Signal]) -> Union[None, int]:
        """analyze.py:283-313 on ``Signal`` objects (pure datetime/float
        comparisons on the host).  ``process_samples`` does not use this: its
        verdicts come from the detect kernel (``rt_record.shadowed``)."""
        for i, other in enumerate(signals):
            if sig.ts > other.ts + other.duration:
                continue
            if sig.ts + sig.duration < other.ts:
                continue
            if other.max > sig.max:
                return i
        return None

    def filter_shadow_signals(self, signals: List[Signal]) -> List[Signal]:
        """analyze.py:315-328 for a caller-supplied list (see ``is_shadow_of``)."""
        verdict = [SignalAnalyzer.is_shadow_of(s, signals) for s in signals]
        return [s for s, v in zip(signals, verdict) if v is None]
