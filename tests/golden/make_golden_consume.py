#!/usr/bin/env python3
"""Generate tests/golden/consume_cases.npz by running the REFERENCE consumers.

Runs only in the build container (needs /root/reference).  ``radiotracking.consume`` imports
``paho.mqtt.client`` and ``cbor2`` (neither installed).  Stand-ins are injected for the import: a
``Client`` that records what is published instead of sending it, and a ``cbor2`` whose ``dumps``
returns nothing -- so the JSON and CSV payloads below are produced by the reference's own code and
the Python standard library (``MQTTConsumer.add`` consume.py:127-160, ``CSVConsumer.add`` :192-199),
while CBOR payloads are NOT captured here (no cbor2 to produce them; tests/test_consume.py checks CBOR
against RFC 8949 by hand-decoded known answers instead).

The file holds the inputs as arrays (so the native formatter can be fed without Python objects) and the
published topic / JSON / CSV strings plus the CSV file contents.

Usage:  TZ=UTC python tests/golden/make_golden_consume.py
"""
import datetime
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

published = []


class _Client:
    def __init__(self, *a, **k):
        pass

    def connect(self, *a, **k):
        pass

    def loop_start(self):
        pass

    def loop_stop(self):
        pass

    def publish(self, topic, payload, qos=0):
        published.append((topic, payload))


paho = types.ModuleType("paho")
paho_mqtt = types.ModuleType("paho.mqtt")
paho_client = types.ModuleType("paho.mqtt.client")
paho_client.Client = _Client
paho.mqtt = paho_mqtt
paho_mqtt.client = paho_client
cbor2 = types.ModuleType("cbor2")
cbor2.dumps = lambda *a, **k: b""
cbor2.CBORTag = lambda tag, value: (tag, value)
sys.modules.update({"paho": paho, "paho.mqtt": paho_mqtt, "paho.mqtt.client": paho_client, "cbor2": cbor2})

import radiotracking  # noqa: E402
import radiotracking.consume as ref_consume  # noqa: E402

EPOCH = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)
US = datetime.timedelta(microseconds=1)


def main():
    rng = np.random.default_rng(5)
    f32 = lambda x: float(np.float32(x))  # noqa: E731 - powers originate as float32
    devices = ["0", "1", "rtl;sdr", 'quo"te', "zwölf☃", "\U0001F4E1x", "line\nbreak", " sp ace ", "a\\b\t"]
    specials = [0.0, -0.0, 1e-5, 9.999e-5, 1e-4, 1e16, 9007199254740993.0, 1.5e300, 5e-324, float("nan"), float("inf"),
                float("-inf"), 100.0, 1 / 3, 123456.789]
    sigs = []
    for i in range(260):
        dev = devices[int(rng.integers(0, 2))] if i < 200 else devices[i % len(devices)]
        ts_us = 1704067200_000000 + int(rng.integers(0, 10**12))
        if i % 7 == 0:
            ts_us -= ts_us % 10**6  # whole second: no fractional part in str()/isoformat()
        if i % 31 == 0:
            ts_us = int(rng.integers(-10**15, 10**15))  # around and before the epoch
        dur_us = int(rng.integers(1, 60000)) if i % 11 else int(rng.integers(0, 3)) * 10**6
        freq = 150.0e6 + float(rng.integers(-2000, 2000)) * 1171.875
        vals = [f32(rng.uniform(-120, -20)) for _ in range(5)]
        if i >= 200:
            vals[int(rng.integers(0, 5))] = specials[i % len(specials)]
            if i % 3 == 0:
                freq = specials[(i // 3) % len(specials)]
        sigs.append((dev, ts_us, dur_us, freq, *vals))

    mq = ref_consume.MQTTConsumer("localhost", 1883, 1, 60, 0, prefix="station/radiotracking")
    sig_file = io.StringIO()
    sig_csv = ref_consume.CSVConsumer(sig_file, cls=radiotracking.Signal, header=radiotracking.Signal.header)
    topics, jsons, csvs = [], [], []
    for dev, ts_us, dur_us, freq, mx, avg, std, noise, snr in sigs:
        s = radiotracking.Signal(dev, EPOCH + ts_us * US, freq, dur_us * US, mx, avg, std, noise, snr)
        published.clear()
        mq.add(s)
        sig_csv.add(s)
        sig_csv.add("not a signal")
        topics.append([t for t, _ in published])
        jsons.append(published[0][1])
        csvs.append(published[1][1])

    # matched signals: groups built through the reference's own add_member
    mdevices = ["0", "1", "2", "we;ird"]
    match_file = io.StringIO()
    match_csv = ref_consume.CSVConsumer(match_file, cls=radiotracking.MatchingSignal,
                                        header=radiotracking.MatchingSignal(mdevices).header)
    m_rows, m_avgs, m_present, m_topics, m_jsons, m_csvs = [], [], [], [], [], []
    for i in range(80):
        g = radiotracking.MatchingSignal(mdevices)
        members = rng.permutation(len(mdevices))[: int(rng.integers(1, len(mdevices) + 1))]
        for d in members:
            ts_us = 1704067200_000000 + i * 10**6 + int(rng.integers(0, 2000)) * (0 if i % 9 == 0 else 1)
            g.add_member(radiotracking.Signal(mdevices[d], EPOCH + ts_us * US, 150.0e6 + float(rng.integers(0, 5)) * 1171.875,
                                              int(rng.integers(8000, 40000)) * US, -40.0, f32(rng.uniform(-90, -30)) if i % 13 else 0.0,
                                              1.0, -100.0, 10.0))
        published.clear()
        mq.add(g)
        match_csv.add(g)
        m_rows.append(((g.ts - EPOCH) // US, g.duration // US, g.frequency))
        m_avgs.append([a if a is not None else np.nan for a in g._avgs])
        m_present.append([a is not None for a in g._avgs])
        m_topics.append([t for t, _ in published])
        m_jsons.append(published[0][1])
        m_csvs.append(published[1][1])

    # state messages (analyze.py:180-190) through the same consumers
    state_file = io.StringIO()
    state_csv = ref_consume.CSVConsumer(state_file, cls=radiotracking.StateMessage, header=radiotracking.StateMessage.header)
    st_in, st_topics, st_json, st_csv, st_repr = [], [], [], [], []
    for i, (dev, state) in enumerate([("0", 2), ("0", 1), ("1", 0), ("rtl;sdr", 1), ("0", 1)]):
        ts_us = 1704067200_000000 + i * 60_000000 + (0 if i == 1 else 123456 * i)
        m = radiotracking.StateMessage(dev, EPOCH + ts_us * US, state)
        published.clear()
        mq.add(m)
        state_csv.add(m)
        st_in.append((dev, ts_us, state))
        st_topics.append([t for t, _ in published])
        st_json.append(published[0][1])
        st_csv.append(published[1][1])
        st_repr.append(repr(m))

    out = dict(
        st_device=np.array([x[0] for x in st_in]), st_ts_us=np.array([x[1] for x in st_in], dtype=np.int64),
        st_state=np.array([x[2] for x in st_in], dtype=np.int32), st_topics=np.array(st_topics), st_json=np.array(st_json),
        st_csv=np.array(st_csv), st_repr=np.array(st_repr), st_csv_file=np.array(state_file.getvalue()),
        devices=np.array(devices), sig_device=np.array([s[0] for s in sigs]),
        sig_ts_us=np.array([s[1] for s in sigs], dtype=np.int64), sig_dur_us=np.array([s[2] for s in sigs], dtype=np.int64),
        sig_freq=np.array([s[3] for s in sigs], dtype=np.float64), sig_vals=np.array([s[4:] for s in sigs], dtype=np.float64),
        sig_topics=np.array(topics), sig_json=np.array(jsons), sig_csv=np.array(csvs), sig_csv_file=np.array(sig_file.getvalue()),
        m_devices=np.array(mdevices), m_ts_us=np.array([r[0] for r in m_rows], dtype=np.int64),
        m_dur_us=np.array([r[1] for r in m_rows], dtype=np.int64), m_freq=np.array([r[2] for r in m_rows], dtype=np.float64),
        m_avgs=np.array(m_avgs, dtype=np.float64), m_present=np.array(m_present, dtype=np.uint8),
        m_topics=np.array(m_topics), m_json=np.array(m_jsons), m_csv=np.array(m_csvs), m_csv_file=np.array(match_file.getvalue()),
    )
    np.savez_compressed(os.path.join(HERE, "consume_cases.npz"), **out)
    print(len(sigs), "signals,", len(m_rows), "matched; sample:")
    print(jsons[0]); print(csvs[0]); print(m_jsons[0]); print(m_csvs[0]); print(topics[0], m_topics[0])


if __name__ == "__main__":
    main()
