#!/usr/bin/env python3
"""Randomised parity soak (GPU box): random geometries, thresholds, durations, calibrations, lanes and pulse
placements (incl. pulses across buffer boundaries and at buffer ends), every stream of every case compared with the
oracle record by record.  Not part of the pytest suite (minutes of runtime); prints one line per case and a
summary.  usage: soak_parity.py [seconds] [seed]"""
import datetime
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import analyze_oracle as oracle  # noqa: E402
from pyradiotracking_amd import synth  # noqa: E402
from pyradiotracking_amd.analyze import BatchSignalAnalyzer  # noqa: E402

TS0 = datetime.datetime(2024, 1, 1)
BIG = os.environ.get("SOAK_BIG") == "1"  # long buffers (1000..8000 segments), 16..128 streams, many pulses per stream
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
n_cases = n_records = n_bad = n_unexplained = n_field = 0
case = int(os.environ.get("SOAK_FIRST_CASE", "1")) - 1  # (SOAK_FIRST_CASE=n: start at case n of the seed -- a case depends on (seed, case number) only)
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng([seed0, case])
    nperseg = int(rng.choice([256, 256, 512, 1024, 2048, 4096]))
    fs = int(rng.choice([300000, 1024000, 2048000, 2400000, 3200000]))
    window = rng.choice(["hamming", "hann", "blackman", "boxcar"])
    n_seg = int(rng.integers(2, 400))
    if BIG:
        nperseg = int(rng.choice([256, 256, 1024]))
        n_seg = int(rng.integers(1000, 8001)) * 256 // nperseg
    blen = n_seg * nperseg + int(rng.integers(0, nperseg))
    n_streams = int(rng.integers(1, 9)) if rng.random() < 0.85 else int(rng.integers(9, 41))
    if BIG:
        n_streams = int(rng.integers(16, 129))
    n_buf = int(rng.integers(2, 4))
    hop = nperseg / fs
    min_ms = float(rng.choice([0.0, 2 * hop * 1e3, 8.0, 5.0]))
    max_ms = float(max(min_ms + 3 * hop * 1e3, rng.choice([10.0, 40.0, 80.0])))
    thr = float(rng.choice([-90.0, -85.0, -100.0]))
    snr = float(rng.choice([5.0, 0.0, 8.0]))
    cal = [float(c) for c in rng.uniform(-6, 6, n_streams)] if rng.random() < 0.5 else 0.0
    mode = str(rng.choice(["sparse", "dense", "auto", "auto", "prefilter", "runfilter"]))  # prefilter: refused where the minimum duration is too short; runfilter: where it does not fit the planning tiles
    # round 5 (SOAK_GENERAL=1, its own random stream so that the cases of earlier rounds stay what they were): a third of the cases at
    # a power of two the fused scans do not cover -- the general transform on the dense path (AUTO / dense only)
    if os.environ.get("SOAK_GENERAL") == "1" and not BIG:
        rng_g = np.random.default_rng([seed0, case, 11])
        only = [int(v) for v in os.environ.get("SOAK_GENERAL_SIZES", "").split(",") if v]  # (e.g. "32,64,128": every case at one of these)
        if only or rng_g.random() < 0.34:
            nperseg = int(rng_g.choice(only or [8, 32, 64, 128, 128, 8192, 8192, 16384, 12, 100, 300, 300, 1000, 1000, 1500, 4099, 6000]))  # (not powers of two: Bluestein)
            n_seg = int(rng_g.integers(2, 400 if nperseg <= 128 else 60))
            blen = n_seg * nperseg + int(rng_g.integers(0, nperseg))
            hop = nperseg / fs
            min_ms = float(rng_g.choice([0.0, 2 * hop * 1e3, 8.0, 5.0]))
            max_ms = float(max(min_ms + 3 * hop * 1e3, rng_g.choice([10.0, 40.0, 80.0])))
            # (round 6: 32 / 64 / 128 / 8192 / 16384 are fused scans -- every mode they have; the other sizes live on the dense path)
            if nperseg not in (32, 64, 128, 8192, 16384):
                mode = "auto" if mode != "dense" else "dense"
            if nperseg >= 4099:
                n_streams = min(n_streams, 6)
                if isinstance(cal, list):
                    cal = cal[:n_streams]
    subtract_first = bool(rng.random() < 0.3)  # SciPy's order of the constant detrend instead of the linearity form
    # round 2: a quarter of the cases with the noise floor around the absolute threshold (8 dB under .. 2 dB over): the
    # sparse path overflows, AUTO climbs to the run-length pre-filter or the dense path; decisions then sit on the noise
    noisy = bool(rng.random() < 0.25)
    if BIG and noisy:
        # the oracle walks every run of a noisy spectrogram in Python: minutes per stream at the BIG sizes
        n_streams = min(n_streams, 16)
        if isinstance(cal, list):
            cal = cal[:n_streams]
        if n_seg > 2000 * 256 // nperseg:
            n_seg = 2000 * 256 // nperseg
            blen = n_seg * nperseg + blen % nperseg
    # ... and in another fifth only one or two of the streams: AUTO re-runs just those dense (n_dense_streams)
    noisy_some = set() if noisy or n_streams < 5 or rng.random() > 0.2 else set(int(x) for x in rng.choice(n_streams, size=int(rng.integers(1, 3)), replace=False))
    lanes = int(rng.choice([1, 1, 2, 3]))
    pipelined = bool(rng.random() < 0.4)   # enqueue buffer k + 1 before fetching buffer k
    vary_len = bool(rng.random() < 0.3)    # shorter buffers than sdr_callback_length
    resets = bool(rng.random() < 0.3)      # single streams restarted between buffers (rt_reset_stream)
    u8 = bool(rng.random() < 0.2)          # RTL-SDR wire format: uint8 I/Q converted in the scan kernel's load
    if u8:
        thr = -70.0
    dev_tensor = bool(rng.random() < 0.3)  # torch CUDA tensors (rows padded: stream_stride > n_samples) instead of host arrays
    chunking = int(rng.choice([0, 0, 4, 8, 16]))  # segments per lane-group chunk (0 = the library's choice)
    w = oracle.window_coefficients(window, nperseg)
    iq = []
    for s in range(n_streams):
        total = n_buf * blen
        pulses = synth.random_pulses(rng, total, fs, w, int(rng.integers(2, 12)) * (n_buf * 8 if BIG else 1), dur_ms=(min(0.5 * max_ms, 4.0), 1.3 * max_ms),
                                     peak_dbw=(thr - 4.0, thr + 30.0))
        for k in range(1, n_buf):  # across a boundary, and one that ends right at a boundary
            amp = synth.amp_for_peak_dbw(thr + 20.0, w, fs)
            ln = int(min(0.6 * max_ms, 12.0) * 1e-3 * fs)
            pulses.append(synth.Pulse(max(0, k * blen - ln // 2), ln, float(rng.uniform(-0.4, 0.4) * fs), amp, 0.1))
            pulses.append(synth.Pulse(max(0, k * blen - ln - int(rng.integers(0, 3)) * nperseg), ln, float(rng.uniform(-0.4, 0.4) * fs), amp, 0.3))
        dc = complex(2e-3, -1e-3) if rng.random() < 0.3 else 0j
        # (round 3: up to 10 dB OVER the threshold -- the exact pre-filter's per-bin thresholds, AUTO's four levels)
        sigma = synth.NOISE_SIGMA if not (noisy or s in noisy_some) else float(np.sqrt(10.0 ** ((thr + rng.uniform(-8.0, 10.0)) / 10.0) * fs / 2.0))
        iq.append(synth.make_stream(synth.StreamSpec(total, fs, pulses, dc=dc, noise_sigma=sigma), 1000 * case + s))
    iq = np.stack(iq)
    # ... and in a third of the noisy cases the level of some or all streams falls by 6 dB at a buffer boundary (thresholds
    # taken from the buffer before are then too high: check_bin_thresholds, the re-run on the call's own row means)
    rng_step = np.random.default_rng([seed0, case, 7])
    floor_step = noisy and rng_step.random() < 0.33
    if floor_step:
        k0 = int(rng_step.integers(1, n_buf))
        who = np.arange(n_streams) if rng_step.random() < 0.5 else rng_step.choice(n_streams, size=max(1, n_streams // 4), replace=False)
        iq[who, k0 * blen:] *= np.complex64(0.5)
    poison = os.environ.get("SOAK_POISON") == "1" and not u8 and rng.random() < 0.5
    if poison:
        # NaN samples (SOAK_POISON_KIND=nan, the default): a NaN segment and NaN row means in every bin, the same on both
        # sides.  Inf / huge samples (SOAK_POISON_KIND=all) overflow inside the FFT: whether a bin ends up Inf or NaN -- and
        # with it whether a row mean is Inf (rejects every other cell of the row) or NaN (accepts them) -- depends on the
        # order of pocketfft's additions, so the reference itself is not a stable yardstick there; that kind only checks
        # that nothing crashes.
        for _ in range(int(rng.integers(1, 4))):
            iq[int(rng.integers(0, n_streams)), int(rng.integers(0, iq.shape[1]))] = rng.choice(
                np.array([np.nan, complex(np.nan, 1.0)] if os.environ.get("SOAK_POISON_KIND", "nan") == "nan"
                         else [np.nan, np.inf, complex(0, -np.inf), 1e30, complex(np.nan, 1.0), 3e38], dtype=np.complex64))
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=window, signal_min_duration_ms=min_ms, signal_max_duration_ms=max_ms,
              signal_threshold_dbw=thr, snr_threshold_db=snr)
    # (round 6: the per-stream record capacity STARTS small -- a stream that needs more grows it inside rt_fetch, so no buffer is
    # skipped for its record count any more; a mode the geometry does not have -- the chunk-bit pre-filter with a short minimum
    # duration, the exact one beyond its planner's counters or at nperseg 32 / 8192 / 16384 -- falls back to AUTO instead of
    # wasting the case)
    # round 6 (its own random stream: the cases of earlier rounds stay what they were): half of the cases with the sparse detection's
    # one-wave-per-stream form forced (detect_group; the library's own rule needs 1 024 streams per handle) -- nperseg <= 256 only
    whole_stream = bool(np.random.default_rng([seed0, case, 13]).random() < 0.5) and nperseg <= 256
    b = None
    for m_try in (mode, "auto"):
        try:
            b = BatchSignalAnalyzer([str(i) for i in range(n_streams)], sdr_callback_length=blen, mode=m_try, lanes=lanes, calibration_db=cal,
                                    record_capacity=64, segs_per_chunk=chunking, subtract_first=subtract_first, group_detect=True if whole_stream else None, **kw)
            mode = m_try
            break
        except Exception as e:
            print(f"case {case}: create failed in mode {m_try}: {e}")
    if b is None:
        continue
    cals = cal if isinstance(cal, list) else [cal] * n_streams
    oas = [oracle.OracleAnalyzer(device=str(s), calibration_db=cals[s], **kw) for s in range(n_streams)]
    bad = 0
    nrec = 0
    # the case as a list of events, then one pass over the GPU (fetches lag one call behind the enqueues when
    # `pipelined`: two calls in flight)
    events = []
    for k in range(n_buf):
        n_k = blen if not vary_len else int(rng.integers(2 * nperseg, blen + 1))
        chunk = np.ascontiguousarray(iq[:, k * blen:k * blen + n_k])
        raw = None
        if u8:
            raw = synth.quantize_u8(chunk, gain=float(rng.choice([300.0, 2000.0])))
            chunk = synth.u8_to_complex64_like_kernel(raw)  # what the kernel makes of the bytes: the oracle's input
        if resets and k > 0:
            events += [("reset", s) for s in range(n_streams) if rng.random() < 0.3]
        events.append(("buffer", chunk, raw))
    in_flight, results = [], []

    def fetch_one():
        chunk_ = in_flight.pop(0)
        try:
            results.append(("records", b.fetch_records(), chunk_))
        except Exception as e:
            results.append(("skipped", str(e), chunk_))

    for ev in events:
        if ev[0] == "reset":
            while in_flight:  # a restart takes effect with the next rt_process: settle what is in flight first
                fetch_one()
            b.reset_stream(ev[1])
            results.append(("reset", ev[1]))
            continue
        if dev_tensor:
            import torch

            pad = int(rng.integers(0, 5)) * 8
            src = ev[2] if u8 else ev[1]
            wide = torch.zeros((src.shape[0], src.shape[1] + (2 * pad if u8 else pad)), dtype=torch.uint8 if u8 else torch.complex64, device="cuda")
            view = wide[:, : src.shape[1]]
            view.copy_(torch.from_numpy(src))
            (b.enqueue_bytes(view) if u8 else b.enqueue(view))
        else:
            (b.enqueue_bytes(ev[2]) if u8 else b.enqueue(ev[1]))
        in_flight.append(ev[1])
        if len(in_flight) > (1 if pipelined else 0):
            fetch_one()
    while in_flight:
        fetch_one()
    k = -1
    abandoned = False
    for res in results:
        if abandoned:
            print(f"case {case}: rest of the case skipped (the reference raises IndexError on this ragged sequence)")
            break
        if res[0] == "reset":
            oas[res[1]].reset()
            continue
        k += 1
        chunk = res[2]
        if res[0] == "skipped":  # (a sparse-mode handle whose candidate lists overflowed: no result by contract)
            print(f"case {case}: buffer {k} skipped: {res[1]}")
            try:
                for s in range(n_streams):
                    oas[s].process(chunk[s], TS0)  # both sides keep this buffer as their look-back
            except IndexError:
                abandoned = True
            continue
        rec = res[1]
        for s in range(n_streams):
            try:
                want, kept = oas[s].process(chunk[s], TS0)
            except IndexError:
                # the reference indexes the CURRENT time axis with a look-back offset (analyze.py:422-423): a run reaching
                # further back than the current buffer has columns raises there (only possible with varying buffer lengths)
                abandoned = True
                break
            mine = rec[rec["stream"] == s]
            nrec += len(want)
            got = [(int(r["fi"]), int(r["start"]), int(r["end"])) for r in mine]
            exp = [(x.fi, x.start, x.end) for x in want]
            if got != exp:
                # decisions that sit on float32 round-off of the powers may flip: report them, with the margin
                bad += 1
                extra, missing = sorted(set(got) - set(exp)), sorted(set(exp) - set(got))
                # margin of the decisions behind the differing runs: the cells at their ends against both thresholds
                spec = oas[s].spec_last  # [F, T] float32 of this buffer (just stored by process())
                thr_lin = np.float32(oracle.db_to_linear(thr + cals[s]))
                snr_lin = np.float32(oracle.db_to_linear(snr))
                margins = []
                for fi, st, en in (extra + missing)[:16]:
                    avg = spec[fi].mean()
                    lo, hi = max(0, st - 1), min(spec.shape[1], en + 2)
                    pw = spec[fi, lo:hi]
                    if len(pw):
                        margins.append(float(np.minimum(np.abs(pw / thr_lin - 1), np.abs(pw / avg / snr_lin - 1)).min()))
                m = min(margins) if margins else float("nan")
                # float32 round-off of the powers reaches 1e-7 x (strongest cell of the segment / this cell) in amplitude:
                # decisions closer than 2e-4 to a threshold are not decidable in float32
                kind = "round-off margin" if m < 2e-4 else "UNEXPLAINED"
                n_unexplained += m >= 2e-4
                print(f"  MISMATCH ({kind}) case {case} buf {k} stream {s}: {len(got)} vs {len(exp)} records; extra {extra[:4]} missing {missing[:4]}; "
                      f"smallest decision margin in the differing runs {m:.2e}")
                continue
            kept_ids = {id(x) for x in kept}
            if [bool(r["shadowed"]) for r in mine] != [id(x) not in kept_ids for x in want]:
                bad += 1
                gv = [bool(r["shadowed"]) for r in mine]
                ov = [id(x) not in kept_ids for x in want]
                i = [a_ != b_ for a_, b_ in zip(gv, ov)].index(True)
                xi = want[i]
                gdb = 10 * np.log10(mine["max_p"].astype(np.float32)) - np.float32(cals[s])
                near = []
                for j, xj in enumerate(want):
                    if j != i and not (xi.ts > xj.ts + xj.duration) and not (xi.ts + xi.duration < xj.ts):
                        near.append((abs(float(xj.max) - float(xi.max)), j, float(xj.max) - float(xi.max), float(gdb[j]) - float(gdb[i])))
                near.sort()
                print(f"  SHADOW MISMATCH case {case} buf {k} stream {s}: record {i} (fi {xi.fi}, {xi.start}..{xi.end}) gpu shadowed={gv[i]} oracle shadowed={ov[i]}; "
                      f"closest overlapping maxima (oracle dB difference, gpu dB difference): {[(round(d, 7), round(g_, 7)) for _, _, d, g_ in near[:3]]}")
                continue
            sigs = b.decoder.signals(mine, [str(i) for i in range(n_streams)], [TS0] * n_streams)
            for g, x in zip(sigs, want):
                ok = g.ts == x.ts and g.duration == x.duration and g.frequency == x.frequency
                for name in ("max", "avg", "noise", "snr", "std"):
                    a_, b_ = getattr(g, name), getattr(x, name)
                    ok = ok and (abs(a_ - b_) < 0.1 or (np.isnan(a_) and np.isnan(b_)))  # the north_star bar
                if not ok:
                    bad += 1
                    n_field += 1
                    print(f"  FIELD MISMATCH case {case} buf {k} stream {s}: gpu max/avg/std/noise/snr {g.max} {g.avg} {g.std} {g.noise} {g.snr} vs {x}")
                    if os.environ.get("SOAK_DUMP_CELLS"):
                        spec = oas[s].spec_last
                        cells = spec[x.fi, max(x.start, 0):x.end]
                        print(f"    oracle cells of the record (this buffer's part) {cells.tolist()}; strongest bin of the head segment {float(spec[:, max(x.start, 0)].max())}")
                        # the samples of the record's segments (complex64 as the oracle saw them), for a fixture
                        np.save(os.path.join(os.environ["SOAK_DUMP_CELLS"], f"case{case}_buf{k}_stream{s}_fi{x.fi}_seg{max(x.start, 0)}_{x.end}.npy"),
                                chunk[s][max(x.start, 0) * nperseg:x.end * nperseg])
                    break
    b.close()
    n_cases += 1
    n_records += nrec
    n_bad += bad
    print(f"case {case}: N={nperseg} fs={fs} {window} T={n_seg} S={n_streams} bufs={n_buf} min/max={min_ms:.2f}/{max_ms:.1f} ms thr={thr} snr={snr} "
          f"mode={mode} lanes={lanes} cal={'per-stream' if isinstance(cal, list) else cal}{' pipelined' if pipelined else ''}{' ragged' if vary_len else ''}{' restarts' if resets else ''}{' uint8' if u8 else ''}{' device-tensors' if dev_tensor else ''}{' poisoned' if poison else ''}{' noisy' if noisy else ''}{' floor-step' if floor_step else ''}{' noisy-streams=' + str(sorted(noisy_some)) if noisy_some else ''}{' subtract-first' if subtract_first else ''}{' wave-per-stream' if whole_stream else ''} chunk={chunking}: {nrec} records, {bad} mismatching stream-buffers", flush=True)
print(f"SOAK: {n_cases} cases, {n_records} oracle records, {n_bad} mismatching stream-buffers "
      f"({n_unexplained} not explained by a float32 round-off margin, {n_field} with a field beyond 0.1 dB)")
