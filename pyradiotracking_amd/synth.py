"""Synthetic SDR streams (there is no SDR hardware on the GPU box).

Recipe of SURVEY.md section 8(d): complex white noise (sigma 1e-5 per
component, far below the -90 dBW detection threshold) plus rectangular tone
pulses ``A*exp(2j*pi*(f*t+phi))``.  A pulse amplitude is chosen from a target
peak PSD via the processing gain of the window,
``P_peak[dBW] = 20*log10(A) + 10*log10((sum w)^2 / (fs * sum w^2))``.

Two generators:

* :func:`make_stream` -- NumPy (PCG64), bit-reproducible from the seed; used
  for fixtures and parity tests.
* :func:`make_batch_device` -- torch, generates ``[S, B]`` complex64 directly
  in HBM for throughput runs (not bit-reproducible across devices; parity on
  such data is always checked by copying sample streams back to the host).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

NOISE_SIGMA = 1e-5


@dataclass
class Pulse:
    start: int  # first sample
    length: int  # samples
    freq: float  # Hz offset from centre
    amp: float
    phase: float = 0.0


@dataclass
class StreamSpec:
    n_samples: int
    sample_rate: float
    pulses: List[Pulse] = field(default_factory=list)
    noise_sigma: float = NOISE_SIGMA
    dc: complex = 0j


def window_gain_db(window: np.ndarray, sample_rate: float) -> float:
    """PSD of a bin-centred unit-amplitude tone, in dB (G)."""
    w = np.asarray(window, dtype=np.float64)
    return 10.0 * math.log10(w.sum() ** 2 / (sample_rate * (w * w).sum()))


def amp_for_peak_dbw(peak_dbw: float, window: np.ndarray, sample_rate: float) -> float:
    return 10.0 ** ((peak_dbw - window_gain_db(window, sample_rate)) / 20.0)


def make_stream(spec: StreamSpec, seed: int) -> np.ndarray:
    """complex64 IQ of one stream.  Noise reals are drawn first, then imags
    (the order the golden-vector script pins)."""
    rng = np.random.default_rng(seed)
    n = spec.n_samples
    x = np.empty(n, dtype=np.complex128)
    x.real = rng.standard_normal(n)
    x.imag = rng.standard_normal(n)
    x *= spec.noise_sigma
    if spec.dc:
        x += spec.dc
    for p in spec.pulses:
        a = max(0, p.start)
        b = min(n, p.start + p.length)
        if b <= a:
            continue
        t = np.arange(a, b, dtype=np.float64) / spec.sample_rate
        x[a:b] += p.amp * np.exp(2j * np.pi * (p.freq * t + p.phase))
    return x.astype(np.complex64)


U8_SCALE = np.float32(1.0 / 127.5)


def quantize_u8(x: np.ndarray, gain: float = 1.0) -> np.ndarray:
    """complex IQ -> the RTL-SDR wire format: interleaved uint8 (I, Q), value = round((v*gain+1)*127.5)
    clipped to 0..255 -- the inverse of pyrtlsdr's ``packed_bytes_to_iq``."""
    z = np.asarray(x, dtype=np.complex128) * gain
    out = np.empty(z.shape + (2,), dtype=np.float64)
    out[..., 0] = z.real
    out[..., 1] = z.imag
    q = np.clip(np.rint((out + 1.0) * 127.5), 0, 255).astype(np.uint8)
    return q.reshape(z.shape[:-1] + (2 * z.shape[-1],))


def u8_to_complex64_like_kernel(raw: np.ndarray) -> np.ndarray:
    """What the scan kernel makes of wire bytes: fma(byte, float32(1/127.5), -1) per component,
    one rounding (evaluated exactly here: the product fits float64)."""
    v = np.asarray(raw, dtype=np.float64) * np.float64(U8_SCALE) - 1.0
    v = v.astype(np.float32)
    return (v[..., 0::2] + 1j * v[..., 1::2]).astype(np.complex64)


def u8_to_complex128_like_pyrtlsdr(raw: np.ndarray) -> np.ndarray:
    """pyrtlsdr ``packed_bytes_to_iq``: bytes -> float64, /= 127.5, -= (1+1j)."""
    v = np.asarray(raw, dtype=np.float64)
    iq = v[..., 0::2] + 1j * v[..., 1::2]
    iq /= 127.5
    iq -= 1 + 1j
    return iq


def random_pulses(
    rng: np.random.Generator,
    n_samples: int,
    sample_rate: float,
    window: np.ndarray,
    n_pulses: int,
    dur_ms: Sequence[float] = (15.0, 15.0),
    peak_dbw: Sequence[float] = (-80.0, -60.0),
    keep_clear_tail: Optional[int] = None,
) -> List[Pulse]:
    """``n_pulses`` pulses with uniform start/frequency/level.  With
    ``keep_clear_tail`` the last that many samples stay pulse-free (so no run
    straddles the buffer end unless a test wants it)."""
    pulses = []
    for _ in range(n_pulses):
        dur = rng.uniform(dur_ms[0], dur_ms[1]) * 1e-3
        length = int(round(dur * sample_rate))
        hi = n_samples - length - (keep_clear_tail or 0)
        start = int(rng.integers(0, max(1, hi)))
        freq = float(rng.uniform(-0.45, 0.45) * sample_rate)
        amp = amp_for_peak_dbw(float(rng.uniform(*peak_dbw)), window, sample_rate)
        pulses.append(Pulse(start, length, freq, amp, float(rng.uniform(0, 1))))
    return pulses


NOISE_BLOCK = 64  # streams per noise generator state (blocks are aligned in GLOBAL stream numbering)


def make_batch_device(
    n_streams: int,
    n_samples: int,
    sample_rate: float,
    window: np.ndarray,
    pulses_per_stream: Sequence[int] = (4, 8),
    dur_ms: Sequence[float] = (15.0, 15.0),
    peak_dbw: Sequence[float] = (-80.0, -60.0),
    seed: int = 0,
    device="cuda",
    noise_sigma: float = NOISE_SIGMA,
    trains: bool = False,
    first_stream: int = 0,
):
    """``[S, B]`` complex64 batch generated in device memory with torch: streams
    ``first_stream .. first_stream + n_streams - 1`` of the population ``seed``.

    A stream's content depends on ``(seed, global stream number)`` only, so a population
    sharded over ranks (``shard.stream_range``) is the same population whatever the number
    of ranks: noise comes from a device ``torch.Generator`` re-seeded per aligned block of
    ``NOISE_BLOCK`` global streams, pulses are drawn on the host (NumPy, seeded per global
    stream) and added on the device, a pulse at a time (they are sparse).  Pulses never
    touch the last segment of a buffer, so nothing straddles the end.
    """
    import torch

    dev = torch.device(device)
    out = torch.empty((n_streams, n_samples), dtype=torch.complex64, device=dev)
    gen = torch.Generator(device=dev)
    real_view = torch.view_as_real(out)  # [S, B, 2] float32
    g0, g1 = first_stream, first_stream + n_streams
    for blk in range(g0 // NOISE_BLOCK, (g1 + NOISE_BLOCK - 1) // NOISE_BLOCK):
        gen.manual_seed((int(seed) * 1000003 + blk) & 0x7FFFFFFFFFFFFFFF)
        b0, b1 = blk * NOISE_BLOCK, (blk + 1) * NOISE_BLOCK
        if b0 >= g0 and b1 <= g1:
            real_view[b0 - g0 : b1 - g0].normal_(0.0, noise_sigma, generator=gen)
        else:  # block cut by the shard boundary: generate it whole, keep the own part
            tmp = torch.empty((NOISE_BLOCK, n_samples, 2), dtype=torch.float32, device=dev)
            tmp.normal_(0.0, noise_sigma, generator=gen)
            lo, hi = max(b0, g0), min(b1, g1)
            real_view[lo - g0 : hi - g0] = tmp[lo - b0 : hi - b0]
            del tmp
    nperseg = len(window)
    plists = []
    for s in range(n_streams):
        rng = np.random.default_rng([seed, first_stream + s])
        k = int(rng.integers(pulses_per_stream[0], pulses_per_stream[1] + 1))
        if trains:
            plists.append(tag_trains(rng, n_samples, sample_rate, window, peak_dbw=peak_dbw, keep_clear_tail=2 * nperseg))
        else:
            plists.append(random_pulses(rng, n_samples, sample_rate, window, k, dur_ms, peak_dbw, keep_clear_tail=2 * nperseg))
    _add_pulses(out, plists, sample_rate)
    return out


def _add_pulses(out, plists, sample_rate, max_elems: int = 1 << 26):
    """Add every stream's pulses to ``out`` ([S, B] complex64, any torch device), a pulse SLOT at a time: the k-th pulse
    of every stream that has one is generated in one batch (rows padded to the slot's longest pulse, at most
    ``max_elems`` padded samples per batch).  Within a batch a stream appears once, and the slots are applied in order,
    so a sample gets its tones added in the order of the stream's pulse list -- bit for bit what one
    ``out[s, a:b] += tone`` per pulse gives, in a few thousand kernel launches instead of millions at 32 768 streams."""
    import torch

    dev = out.device
    n_samples = out.shape[1]
    flat = out.view(-1)
    two_pi = 2.0 * math.pi
    kmax = max((len(pl) for pl in plists), default=0)
    for k in range(kmax):
        rows = [s for s, pl in enumerate(plists) if len(pl) > k and pl[k].length > 0]
        i0 = 0
        while i0 < len(rows):
            # (rows of one batch: as many as fit max_elems at the longest pulse among them)
            i1, longest = i0, 0
            while i1 < len(rows):
                ln = max(longest, plists[rows[i1]][k].length)
                if i1 > i0 and ln * (i1 - i0 + 1) > max_elems:
                    break
                longest = ln
                i1 += 1
            part = rows[i0:i1]
            i0 = i1
            ps = [plists[s][k] for s in part]
            col = lambda vals, dt: torch.tensor(vals, dtype=dt, device=dev).unsqueeze(1)
            start = col([p.start for p in ps], torch.int64)
            length = col([p.length for p in ps], torch.int64)
            freq = col([p.freq for p in ps], torch.float64)
            phase = col([p.phase for p in ps], torch.float64)
            amp = col([p.amp for p in ps], torch.float64).to(torch.complex64)
            row = col(part, torch.int64)
            j = torch.arange(longest, device=dev, dtype=torch.int64).unsqueeze(0)
            t = (start + j).to(torch.float64)
            ph = two_pi * (freq * t / sample_rate + phase)
            tone = torch.complex(torch.cos(ph), torch.sin(ph)).to(torch.complex64) * amp
            del ph, t
            keep = j < length
            idx = (row * n_samples + start + j)[keep]
            flat[idx] = flat[idx] + tone[keep]
            del tone, idx, keep


def tag_trains(rng: np.random.Generator, n_samples: int, sample_rate: float, window: np.ndarray,
               n_tags: Sequence[int] = (8, 16), dur_ms: Sequence[float] = (10.0, 38.0),
               period_s: Sequence[float] = (0.1, 1.0), peak_dbw: Sequence[float] = (-80.0, -60.0),
               keep_clear_tail: int = 0) -> List[Pulse]:
    """BASELINE config 5: several tags per stream, each with its own frequency, pulse length,
    level and repetition period; pulses of different tags overlap in time (shadow filter)."""
    pulses: List[Pulse] = []
    total_s = n_samples / sample_rate
    for _ in range(int(rng.integers(n_tags[0], n_tags[1] + 1))):
        freq = float(rng.uniform(-0.45, 0.45) * sample_rate)
        length = int(round(rng.uniform(*dur_ms) * 1e-3 * sample_rate))
        period = float(rng.uniform(*period_s))
        amp = amp_for_peak_dbw(float(rng.uniform(*peak_dbw)), window, sample_rate)
        t = float(rng.uniform(0, period))
        while t < total_s:
            start = int(t * sample_rate)
            if start + length <= n_samples - keep_clear_tail:
                pulses.append(Pulse(start, length, freq, amp, float(rng.uniform(0, 1))))
            t += period
    return pulses


def quantize_u8_device(iq, gain: float = 1.0):
    """``[S, B]`` complex64 CUDA tensor -> ``[S, 2*B]`` uint8 in the RTL-SDR wire format (on device)."""
    import torch

    v = torch.view_as_real(iq)  # [S, B, 2]
    q = torch.clamp(torch.round((v * gain + 1.0) * 127.5), 0, 255).to(torch.uint8)
    return q.reshape(iq.shape[0], 2 * iq.shape[1]).contiguous()
