#!/usr/bin/env python3
"""Randomised soak of the native record serialisers (rt_format_*, host code) against the imported reference
consumers (radiotracking.consume.MQTTConsumer.add; build container only): random float bit patterns, time
stamps incl. pre-epoch and whole seconds, durations, device names with separators / quotes / non-BMP characters;
JSON and CSV payloads and topics compared byte for byte (CBOR cannot be produced here: no cbor2).
usage: soak_consume.py [seconds] [seed]"""
import datetime
import os
import struct
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "/root/reference")
published = []


class _Client:
    def __init__(self, *a, **k):
        pass

    def connect(self, *a, **k):
        pass

    def loop_start(self):
        pass

    def publish(self, topic, payload, qos=0):
        published.append((topic, payload))


paho, paho_mqtt, paho_client, cbor2 = (types.ModuleType(n) for n in ("paho", "paho.mqtt", "paho.mqtt.client", "cbor2"))
paho_client.Client = _Client
paho.mqtt, paho_mqtt.client = paho_mqtt, paho_client
cbor2.dumps = lambda *a, **k: b""
cbor2.CBORTag = lambda tag, value: (tag, value)
sys.modules.update({"paho": paho, "paho.mqtt": paho_mqtt, "paho.mqtt.client": paho_client, "cbor2": cbor2})
import radiotracking  # noqa: E402
import radiotracking.consume as ref_consume  # noqa: E402

import pyradiotracking_amd as mine  # noqa: E402
from pyradiotracking_amd import consume as rtc  # noqa: E402

EPOCH = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)
US = datetime.timedelta(microseconds=1)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
alphabet = list("01ab;,\"' \n\t\\/äß☃") + ["\U0001F4E1", " ", "\x7f", "\x00"[:0] or "x"]
mq = ref_consume.MQTTConsumer("localhost", 1883, 1, 60, 0, prefix="st/rt")
t_end = time.time() + budget
n = bad = 0


def rnd_float():
    k = rng.integers(0, 6)
    if k == 0:
        return float(np.float32(rng.uniform(-120, -20)))
    if k == 1:
        return struct.unpack("<d", struct.pack("<Q", int(rng.integers(0, 2**63)) | (int(rng.integers(0, 2)) << 63)))[0]  # any bit pattern
    if k == 2:
        return float(rng.choice([0.0, -0.0, 1e-4, 9.9999e-5, 1e16, 9.999999999999998e15, 5e-324, float("inf"), float("-inf"), float("nan")]))
    if k == 3:
        return float(rng.integers(-10**6, 10**6))
    if k == 4:
        return float(rng.uniform(-1, 1) * 10.0 ** int(rng.integers(-30, 30)))
    return float(np.float32(rng.uniform(-1, 1)))


while time.time() < t_end:
    dev = "".join(rng.choice(alphabet, size=int(rng.integers(1, 8))))
    ts_us = int(rng.choice([rng.integers(-10**15, 4 * 10**15), 1704067200_000000 + int(rng.integers(0, 10**9)) * 10**6]))
    dur_us = int(rng.choice([0, 1, rng.integers(0, 10**8), int(rng.integers(0, 100)) * 10**6]))
    vals = [rnd_float() for _ in range(6)]
    ts = EPOCH + ts_us * US
    r = radiotracking.Signal(dev, ts, vals[0], dur_us * US, *vals[1:])
    m = mine.Signal(dev, ts, vals[0], dur_us * US, *vals[1:])
    published.clear()
    mq.add(r)
    got = rtc.mqtt_messages(m, prefix="st/rt")
    want = [(t, p) for t, p in published]
    ok = [g[0] for g in got] == [w[0] for w in want] and got[0][1] == want[0][1] and got[1][1] == want[1][1]
    n += 1
    if not ok:
        bad += 1
        print(f"MISMATCH dev {dev!r} ts_us {ts_us} dur_us {dur_us} vals {vals}\n   got  {got[:2]}\n   want {want[:2]}", flush=True)
print(f"SOAK CONSUME: {n} signals, {bad} mismatching payload sets (topic, JSON, CSV)")

# matched groups and state messages through the same consumers
t_end = time.time() + budget / 3
n = bad = 0
while time.time() < t_end:
    n_dev = int(rng.integers(1, 9))
    devs = ["".join(rng.choice(alphabet, size=int(rng.integers(1, 5)))) + str(i) for i in range(n_dev)]
    rg, mg = radiotracking.MatchingSignal(devs), mine.MatchingSignal(devs)
    for d in rng.permutation(n_dev)[: int(rng.integers(1, n_dev + 1))]:
        ts = EPOCH + int(rng.integers(0, 4 * 10**15)) * US
        args = (devs[d], ts, rnd_float(), int(rng.integers(0, 10**7)) * US, rnd_float(), rnd_float(), 1.0, -100.0, 10.0)
        rg.add_member(radiotracking.Signal(*args))
        mg.add_member(mine.Signal(*args))
    st_args = (devs[0], EPOCH + int(rng.integers(-10**14, 4 * 10**15)) * US, int(rng.integers(0, 3)))
    for r, m in ((rg, mg), (radiotracking.StateMessage(*st_args), mine.StateMessage(*st_args))):
        published.clear()
        mq.add(r)
        got = rtc.mqtt_messages(m, prefix="st/rt")
        want = list(published)
        n += 1
        if not ([g[0] for g in got] == [w[0] for w in want] and got[0][1] == want[0][1] and got[1][1] == want[1][1]):
            bad += 1
            print(f"MISMATCH {type(m).__name__}\n   got  {got[:2]}\n   want {want[:2]}", flush=True)
print(f"SOAK CONSUME: {n} matched groups / state messages, {bad} mismatching payload sets")
