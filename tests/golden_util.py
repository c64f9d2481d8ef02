"""Loading helpers for tests/golden/*.npz (data written by make_golden.py)."""
import datetime
import hashlib
import json
import os

import numpy as np

from pyradiotracking_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TS0 = datetime.datetime(2024, 1, 1, 0, 0, 0)
TS0_UTC = TS0.replace(tzinfo=datetime.timezone.utc)

_cache = {}


def _npz(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
    return _cache[name]


def iq_index():
    return json.loads(str(_npz("iq_cases.npz")["index_json"]))


def iq_case_names():
    return [m["name"] for m in iq_index()]


def iq_case(name):
    """-> (meta, buffers, ts_starts, expected) with expected[b] = dict(table, kept, reprs, rows?)."""
    z = _npz("iq_cases.npz")
    meta = next(m for m in iq_index() if m["name"] == name)
    sp = meta["spec"]
    spec = synth.StreamSpec(
        n_samples=sp["n_samples"],
        sample_rate=sp["sample_rate"],
        pulses=[synth.Pulse(int(a), int(b), float(c), float(d), float(e)) for a, b, c, d, e in sp["pulses"]],
        noise_sigma=sp.get("noise_sigma", synth.NOISE_SIGMA),
        dc=complex(*sp["dc"]) if "dc" in sp else 0j,
    )
    iq = synth.make_stream(spec, meta["seed"])
    digest = hashlib.sha256(iq.tobytes()).hexdigest()
    assert digest == meta["iq_sha256"], f"regenerated IQ of {name} differs from the pinned bytes"
    blen = meta["buffer_len"]
    buffers = [iq[i * blen : (i + 1) * blen] for i in range(meta["n_buffers"])]
    ts_starts = [TS0 + datetime.timedelta(microseconds=o) for o in meta["ts_offsets_us"]]
    expected = []
    for b in range(meta["n_buffers"]):
        e = dict(
            table=z[f"{name}/b{b}/table"],
            kept=z[f"{name}/b{b}/kept"],
            reprs=[str(s) for s in z[f"{name}/b{b}/reprs"]],
            spec_shape=tuple(z[f"{name}/b{b}/spec_shape"]),
            row_means=z[f"{name}/b{b}/row_means"],
        )
        if f"{name}/b{b}/rows" in z:
            e["rows"] = z[f"{name}/b{b}/rows"]
        expected.append(e)
    kwargs = dict(meta["kwargs"])
    if isinstance(kwargs.get("fft_window"), list):
        w = kwargs["fft_window"]
        kwargs["fft_window"] = tuple(w) if isinstance(w[0], str) else np.asarray(w, dtype=np.float64)
    return meta, kwargs, buffers, ts_starts, expected


def extract_index():
    return json.loads(str(_npz("extract_cases.npz")["index_json"]))


def extract_case(i):
    z = _npz("extract_cases.npz")
    meta = extract_index()[i]
    d = {k: z[f"c{i}/{k}"] for k in ("cur", "last", "freqs", "times", "table", "kept")}
    d["kwargs"] = meta["kwargs"]
    d["has_last"] = meta["has_last"]
    return d


def us(delta: datetime.timedelta) -> int:
    return delta.days * 86400 * 1000000 + delta.seconds * 1000000 + delta.microseconds


def signals_table(signals, ts_start_utc):
    """Same 8 columns as make_golden.signals_to_table, from any objects with
    ts/frequency/duration/max/avg/std/noise/snr attributes."""
    tab = np.zeros((len(signals), 8), dtype=np.float64)
    for i, s in enumerate(signals):
        tab[i] = [us(s.ts - ts_start_utc), s.frequency, us(s.duration), s.max, s.avg, s.std, s.noise, s.snr]
    return tab
