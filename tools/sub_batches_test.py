import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from pyradiotracking_amd import synth
from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients
S, fs, nperseg = 256, 2048000, 256
blen = fs
win = window_coefficients("hamming", nperseg)
iq = synth.make_batch_device(S, blen, fs, win, seed=1000, device="cuda:0")
def make(n_sub):
    ans = []
    per = S // n_sub
    for k in range(n_sub):
        st = torch.cuda.Stream()
        ans.append((BatchSignalAnalyzer([str(i) for i in range(per)], sdr_callback_length=blen, sample_rate=fs, fft_nperseg=nperseg,
                                        mode="auto", timing=True, hip_stream=st.cuda_stream), iq[k * per:(k + 1) * per], st))
    return ans
KMS = [0.0, 0]
def run(ans, n):
    for a, x, st in ans: a.enqueue(x)
    for i in range(n):
        if i + 1 < n:
            for a, x, st in ans: a.enqueue(x)
        for a, x, st in ans:
            a.fetch_records()
            KMS[0] += a.native.call_info().ms_stft; KMS[1] += 1
torch.cuda.synchronize()
for n_sub in (1, 2, 4, 1, 2, 4):
    ans = make(n_sub)
    run(ans, 40); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(ans, 200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("   mean scan launch", round(KMS[0] / KMS[1], 4), "ms for", S // n_sub, "streams ->", round(S // n_sub * 8000 * 256 * 8 / (KMS[0] / KMS[1] * 1e-3) / 1e9), "GB/s per launch"); KMS[0] = 0.0; KMS[1] = 0
    print(n_sub, "sub-batches:", round(S * 8000 * 256 * 200 / dt / 1e6), "MS/s", round(dt / 200 * 1e3, 4), "ms/step")
    del ans
