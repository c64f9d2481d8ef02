import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TZ"] = "UTC"
import numpy as np
from tests.test_gpu_parity import _noisy_batch, _batch_for
fs, nperseg, blen, n_streams = 2048000, 256, 256 * 1500, 6
for thr in (-160.0, -162.0, -163.0):
    iq = _noisy_batch(n_streams, blen, fs, nperseg, seed=int(-thr))
    for cap in (2048, 8192):
        b = _batch_for(dict(sample_rate=fs, signal_threshold_dbw=thr), n_streams, blen, "prefilter", hot_capacity=cap)
        try:
            b.enqueue(np.ascontiguousarray(iq[:, 0])); rec = b.fetch_records()
            info = b.native.call_info()
            print(thr, cap, "n_hot", info.n_hot, "records", len(rec))
        except Exception as e:
            print(thr, cap, "ERR", e)
