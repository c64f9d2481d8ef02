"""Runner integration of the batched analysis path.

The reference's ``Runner`` (``radiotracking/__main__.py:94-190``) starts one ``SignalAnalyzer`` *process*
per ``--device`` entry, pairs it with its ``--calibration`` value, watches ``last_data_ts`` of each and
replaces an analyzer that died or timed out by a fresh one until ``--sdr-max-restart`` is used up.  Here
the SDRs of a station are *streams of a batch*: ``--device`` entries are mapped to (GPU, stream slot)
pairs, one :class:`~pyradiotracking_amd.analyze.BatchSignalAnalyzer` per GPU analyses its subset, and the
life-cycle rules are kept per stream:

===============================  ==========================================================================
reference                        here
===============================  ==========================================================================
``create_and_start`` (:94-129)   :meth:`BatchRunner.start_analyzers` -- slot + calibration per device
callback head (analyze.py        :meth:`BatchRunner.process` -- STARTED / RUNNING heartbeats, running clock,
:204-231)                        drift check, per stream, with the reference's expressions
``check_analyzers`` (:153-190)   :meth:`BatchRunner.check_analyzers` -- time-out -> STOPPED message, restart
                                 budget, *terminate* when it is used up
restart = new process            ``rt_reset_stream`` (the stream loses its look-back, like a fresh
                                 ``_spectrogram_last = None``) + fresh clock; other streams are untouched
``stop_analyzers`` (:143-151)    :meth:`BatchRunner.stop_analyzers`
===============================  ==========================================================================

Not here (out of scope): argument/config parsing, schedules, the SDR I/O itself, MQTT,
dashboard -- the caller owns the producers and hands over one buffer per SDR and step.
"""
from __future__ import annotations

import datetime
import logging
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import pytz

from . import StateMessage
from .shard import stream_range

logger = logging.getLogger(__name__)


def plan_devices(devices: Sequence[str], calibration: Sequence[float], gpus: Sequence[int]) -> List[Tuple[int, int]]:
    """``--device`` / ``--calibration`` lists -> ``(gpu, slot)`` per device: contiguous blocks of devices per
    GPU (sizes differ by at most one, :func:`pyradiotracking_amd.shard.stream_range`).  The calibration
    list must match the device list, as the reference insists (``__main__.py:215-221``)."""
    if len(calibration) != len(devices):
        raise ValueError(f"Calibration values {list(calibration)} do not match devices {list(devices)}.")
    if not gpus:
        raise ValueError("at least one GPU is needed: there is no CPU analysis path")
    plan: List[Tuple[int, int]] = []
    for rank, gpu in enumerate(gpus):
        lo, hi = stream_range(rank, len(gpus), len(devices))
        plan += [(gpu, i - lo) for i in range(lo, hi)]
    return plan


@dataclass
class StreamState:
    """What the reference keeps per analyzer process (analyze.py:84-129) for one stream of a batch."""

    device: str
    calibration_db: float
    gpu: int
    slot: int
    sdr_max_restart: int
    last_data_ts: float = 0.0                        # multiprocessing.Value("d") of the reference (:208-214)
    ts: Optional[datetime.datetime] = None           # `_ts`, the running sample clock (:218-221)
    last_state: Optional[StateMessage] = None
    alive: bool = True                               # False after a fatal clock drift until the restart
    stale: bool = False                              # missed a step: its look-back no longer precedes its next buffer
    restarts: int = field(default=0)


class BatchRunner:
    """Life cycle of the SDR streams of one station around the batched analysis path."""

    def __init__(
        self,
        device: Sequence[str] = ("0",),
        calibration: Sequence[float] = (),
        gpus: Sequence[int] = (0,),
        sdr_max_restart: int = 3,
        sdr_timeout_s: float = 2,
        state_update_s: float = 300,
        signal_queue=None,
        analyzer_factory: Optional[Callable] = None,
        clock: Callable[[], float] = time.time,
        **analysis_kwargs,
    ):
        """``device`` / ``calibration`` / ``sdr_*`` / ``state_update_s`` are the reference's command-line
        arguments of the same names (``__main__.py:44-56``); ``analysis_kwargs`` (sample_rate, fft_nperseg,
        thresholds, ...) go to every analyzer like ``**vars(dargs)`` (:113-118).  ``analyzer_factory(devices,
        calibration_db=[...], gpu=..., **analysis_kwargs)`` defaults to ``BatchSignalAnalyzer``; ``clock``
        returns seconds since the epoch (tests pass their own)."""
        self.devices = [str(d) for d in device]
        calibration = list(calibration)
        if len(calibration) == 0:  # __main__.py:215-217
            calibration = [0.0] * len(self.devices)
            logger.info(f"No calibration values supplied, using {calibration}")
        self.plan = plan_devices(self.devices, calibration, list(gpus))
        self.calibration = [float(c) for c in calibration]
        self.gpus = list(gpus)
        self.sdr_max_restart = sdr_max_restart
        self.sdr_timeout_s = sdr_timeout_s
        self.state_update_s = state_update_s
        self.signal_queue = signal_queue
        self.analysis_kwargs = analysis_kwargs
        self.sample_rate = analysis_kwargs.get("sample_rate", 300000)
        if analyzer_factory is None:
            from .analyze import BatchSignalAnalyzer

            analyzer_factory = BatchSignalAnalyzer
        self._factory = analyzer_factory
        self._clock = clock
        self.running = True
        self.streams: List[StreamState] = []
        self.analyzers: Dict[int, object] = {}   # gpu -> batch analyzer
        self._members: Dict[int, List[int]] = {}  # gpu -> global stream indices in slot order

    # -- start / stop (``__main__.py:94-151``) ---------------------------------------------------------
    def start_analyzers(self):
        if self.analyzers:
            logger.critical("analyzers are already running")
            return
        logger.info("Starting all analyzers")
        self.streams = [
            StreamState(d, c, gpu, slot, self.sdr_max_restart)
            for d, c, (gpu, slot) in zip(self.devices, self.calibration, self.plan)
        ]
        self._members = {}
        for i, st in enumerate(self.streams):
            self._members.setdefault(st.gpu, []).append(i)
        for gpu, members in self._members.items():
            self.analyzers[gpu] = self._factory(
                [self.streams[i].device for i in members],
                calibration_db=[self.streams[i].calibration_db for i in members],
                gpu=gpu,
                **self.analysis_kwargs,
            )
            for i in members:
                logger.info(f"SDR {self.streams[i].device} -> GPU {gpu}, stream slot {self.streams[i].slot}")

    def stop_analyzers(self):
        logger.info("Stopping all analyzers")
        for st in self.streams:  # __main__.py:147
            self._put(StateMessage(st.device, self._from_timestamp(st.last_data_ts), StateMessage.State.STOPPED))
        for an in self.analyzers.values():
            close = getattr(an, "close", None)
            if close:
                close()
        self.analyzers = {}
        self.streams = []

    def terminate(self):
        """``Runner.terminate`` (:192-205) without the signal plumbing."""
        logger.warning(f"terminating {len(self.streams)} analyzers.")
        self.running = False
        self.stop_analyzers()

    # -- helpers ---------------------------------------------------------------------------------------
    def _put(self, msg):
        if self.signal_queue is not None:
            self.signal_queue.put(msg)

    @staticmethod
    def _from_timestamp(ts: float) -> datetime.datetime:
        return datetime.datetime.fromtimestamp(ts, tz=pytz.utc)

    def _update_state(self, st: StreamState, ts: datetime.datetime, state) -> None:
        """analyze.py:180-190."""
        ts = ts.astimezone(pytz.utc)
        last = st.last_state
        if last and last.state == state and last.ts + datetime.timedelta(seconds=self.state_update_s) >= ts:
            return
        st.last_state = StateMessage(st.device, ts, state)
        self._put(st.last_state)

    def _clock_head(self, st: StreamState, n_samples: int, now: float) -> Optional[datetime.datetime]:
        """Head of the reference callback for one stream (analyze.py:204-231).  Returns ``ts_start`` of the
        buffer, or None when the clock drift is fatal (the reference then cancels the SDR's reads and the
        process ends; here the stream waits for ``check_analyzers`` to restart it)."""
        ts_recv = datetime.datetime.fromtimestamp(now)
        buffer_len_dt = datetime.timedelta(seconds=n_samples / self.sample_rate)
        if not st.last_data_ts:
            self._update_state(st, ts_recv, StateMessage.State.STARTED)
        else:
            self._update_state(st, ts_recv, StateMessage.State.RUNNING)
        st.last_data_ts = datetime.datetime.timestamp(ts_recv)
        if not st.ts:
            st.ts = ts_recv
        else:
            st.ts += buffer_len_dt
        clock_drift = (ts_recv - st.ts).total_seconds()
        if clock_drift > 2 * buffer_len_dt.total_seconds():
            logger.warning(
                f"SDR {st.device} total clock drift ({clock_drift:.5f} s) is larger than two blocks, "
                "signal detection is degraded. Terminating..."
            )
            self._update_state(st, ts_recv, StateMessage.State.STOPPED)
            st.alive = False
        return st.ts - buffer_len_dt

    # -- one step: one buffer per SDR ------------------------------------------------------------------
    def process(self, buffers, present: Optional[Sequence[bool]] = None, now: Optional[float] = None):
        """The batched callback: ``buffers`` is ``[n_devices, B]`` complex64 (host array; rows in
        ``--device`` order) or a dict ``gpu -> [S_gpu, B]`` device tensor; ``present[i]`` is False for an SDR
        that delivered nothing this step (its row is ignored).  Signals go to ``signal_queue`` per stream in
        the reference's order, after the shadow filter; returns their number."""
        if not self.analyzers:
            raise RuntimeError("start_analyzers() first")
        now = self._clock() if now is None else now
        n = len(self.streams)
        present = [True] * n if present is None else [bool(p) for p in present]
        if len(present) != n:
            raise ValueError("one `present` flag per device")
        per_gpu = buffers if isinstance(buffers, dict) else None
        host = None if per_gpu is not None else np.asarray(buffers)
        if host is not None and (host.ndim != 2 or host.shape[0] != n):
            raise ValueError(f"expected [{n}, B] buffers")

        # which SDRs take part in this step (decided before anything is enqueued; the clocks move afterwards)
        active = [bool(present[i] and st.alive) for i, st in enumerate(self.streams)]
        resets = []
        for i, st in enumerate(self.streams):
            if not active[i]:
                continue
            if st.stale:
                # the stream missed a step: what the handle holds as "previous buffer" is not the one before
                # this buffer, so no look-back (a gap in the samples; the reference would have hit its
                # clock-drift exit or a restart on the way)
                self.analyzers[st.gpu].reset_stream(st.slot)
                resets.append(i)

        # enqueue on every GPU first (asynchronous), then collect.  If one GPU refuses its call, the calls
        # already enqueued on the others are drained and dropped: rt_fetch is FIFO, a call left pending would
        # pair every later step with the records of the step before it.  Nothing else has changed by then --
        # the stream clocks and heartbeats only advance once every GPU has accepted the step.
        enqueued = []
        try:
            for gpu, members in self._members.items():
                if per_gpu is not None:
                    self.analyzers[gpu].enqueue(per_gpu[gpu])
                else:
                    chunk = np.ascontiguousarray(host[members], dtype=np.complex64)
                    for k, i in enumerate(members):
                        if not active[i]:
                            chunk[k] = 0  # no samples: zero power, below every threshold
                    self.analyzers[gpu].enqueue(chunk)
                enqueued.append(gpu)
        except Exception:
            for gpu in enqueued:
                try:
                    self.analyzers[gpu].fetch_records(allow_truncated=True)
                except Exception as e:  # the step is lost anyway; keep draining the other GPUs
                    logger.error(f"GPU {gpu}: dropping the enqueued step failed: {e}")
            # The step is lost on every GPU.  On the ones that accepted it the handles now hold the dropped buffer as
            # "previous buffer" (a retry of the same buffer would look back into itself), on the refusing one the buffer
            # before it (a gap): no stream may look back from its next buffer -- each starts it with reset_stream.
            for st in self.streams:
                st.stale = True
            raise

        ts_starts: List[Optional[datetime.datetime]] = [None] * n
        for i, st in enumerate(self.streams):
            if not active[i]:
                st.stale = True
                continue
            st.stale = False
            n_samples = host.shape[1] if host is not None else int(per_gpu[st.gpu].shape[1])
            ts_starts[i] = self._clock_head(st, n_samples, now)

        n_signals = 0
        first_error = None
        for gpu, members in self._members.items():
            an = self.analyzers[gpu]
            try:
                # the reference has no limit on signals per buffer; a stream beyond record_capacity keeps
                # record_capacity of its signals instead of costing the whole station this step
                rec = an.fetch_records(allow_truncated=True)
            except Exception as e:  # every GPU is fetched whatever happens on one of them (FIFO, see above)
                logger.error(f"GPU {gpu}: no records for this step: {e}")
                first_error = first_error or e
                continue
            if getattr(getattr(an, "native", None), "last_truncated", False):
                logger.warning(f"GPU {gpu}: a stream exceeded record_capacity; its signal list for this buffer is truncated")
            rec = rec[rec["shadowed"] == 0]
            keep = np.array([ts_starts[members[s]] is not None for s in rec["stream"]], dtype=bool)
            rec = rec[keep]
            names = [self.streams[i].device for i in members]
            starts = [ts_starts[i] for i in members]
            for sig in an.decoder.signals(rec, names, starts):
                self._put(sig)  # analyze.py:251, 280
                n_signals += 1
        if first_error is not None:
            raise first_error
        return n_signals

    # -- liveness (``__main__.py:153-190``) ------------------------------------------------------------
    def check_analyzers(self, now: Optional[float] = None):
        now = self._clock() if now is None else now
        for st in list(self.streams):
            if st.alive:
                if st.last_data_ts == 0.0:  # has not started yet (:163)
                    continue
                if st.last_data_ts > now - self.sdr_timeout_s:  # (:167)
                    continue
                logger.warning(f"SDR {st.device} received last data {datetime.datetime.fromtimestamp(st.last_data_ts)}; timed out.")
                self._put(StateMessage(st.device, self._from_timestamp(st.last_data_ts), StateMessage.State.STOPPED))
            else:
                logger.info(f"SDR {st.device} stream is dead.")
            if st.sdr_max_restart <= 0:  # (:180-183)
                logger.critical(f"SDR {st.device} is dead and beyond restart count, terminating.")
                self.terminate()
                break
            logger.warning(f"Restarting SDR {st.device}.")
            self.restart_stream(st)

    def restart_stream(self, st: StreamState):
        """``create_and_start(device, calibration_db, sdr_max_restart - 1)`` (:186-190) for a stream: fresh
        analyzer state, one restart less, same slot and calibration."""
        an = self.analyzers.get(st.gpu)
        if an is not None:
            an.reset_stream(st.slot)
        st.sdr_max_restart -= 1
        st.restarts += 1
        st.last_data_ts = 0.0
        st.ts = None
        st.last_state = None
        st.alive = True
        st.stale = False
