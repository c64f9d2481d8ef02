"""How full do the candidate lists of a config-5 stream get?  Cells >= threshold (plus the cell before each) per
(stream, bucket = bin & 15), from the oracle-free NumPy spectrogram of sampled streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.fft, scipy.signal
from pyradiotracking_amd import synth
fs, nperseg, blen, S = 3200000, 4096, 3200000, 64
w = scipy.signal.get_window("hamming", nperseg)
iq = synth.make_batch_device(S, blen, fs, w, seed=77, trains=True)
thr = 10 ** (-90 / 10)
w32 = w.astype(np.complex64); scale = 1.0 / (fs * (w32 * w32).sum())
worst = []
for s in range(S):
    x = iq[s].cpu().numpy()
    T = blen // nperseg
    seg = x[: T * nperseg].reshape(T, nperseg)
    seg = seg - seg.mean(axis=-1, keepdims=True)
    P = (np.abs(scipy.fft.fft(w32 * seg)) ** 2 * scale).real.astype(np.float32)  # [T, F]
    hot = P >= thr
    emit = hot.copy(); emit[:-1] |= hot[1:]
    per_bin = emit.sum(axis=0)
    per_bucket = per_bin.reshape(-1, 16).sum(axis=0)
    rng = np.random.default_rng([77, s])
    worst.append((int(per_bucket.max()), s, int(per_bin.max()), int((per_bin > 0).sum()), int(hot.sum())))
worst.sort(reverse=True)
for row in worst[:10]:
    print("max cells in a bucket %d  stream %d  max per bin %d  bins with cells %d  hot cells %d" % row)
