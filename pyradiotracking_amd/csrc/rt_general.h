// rt_general.h -- the spectrogram for any power-of-two nperseg the fused scans do not cover.
//
// The reference hands `fft_nperseg` straight to scipy.signal.spectrogram (radiotracking/__main__.py:59,
// analyze.py:234-241): any integer.  The fused scans (rt_kernels.h, rt_scan64.h) exist for 256 .. 4096; every other
// power of two from 8 to 16 384 (128 and 8 192 are plausible station settings) is served here, on the dense path:
//   stft_general   x -> constant detrend -> window -> FFT -> |X|^2 * scale   (scipy _spectral_py.py:2185-2202, 2126-2128),
//                  written as the dense spectrogram [S][T][N] (+ the look-back tail of the last K segments),
// followed by detect_dense (the extractor on a dense map, any number of bins).  Two passes over 4 bytes per cell like every
// dense call: bound by its 16 bytes per sample, not a path the roofline is quoted on.
//
// Transform: radix-2 decimation in time inside LDS.  A workgroup of 256 threads holds SPB = max(1, 512 / N) segments;
// samples are stored at bit-reversed places, log2 N butterfly passes follow in place (twiddles W_N^m, m < N / 2, from a
// table made in double precision), natural order comes out.  One barrier per pass.
#ifndef RT_GENERAL_H
#define RT_GENERAL_H

#include "rt_kernels.h"

namespace rt {

struct GeneralParams {
    const void *iq;          // [S][stream_stride] complex64, or interleaved uint8 I/Q
    int64_t stream_stride;   // samples
    int32_t n_streams;
    int32_t n_seg;           // T
    int32_t nperseg;         // N, a power of two
    int32_t log2n;
    int32_t segs_per_block;  // SPB
    int32_t tail_cols;       // K
    const float *window;     // [N] window coefficients times sqrt(scale)
    const cf *tw;            // [N / 2] W_N^m
    float *spec;             // [S][T][N]
    float *tail;             // [S][K][N], or null
};

constexpr int kGeneralBlock = 256;
constexpr int kGeneralMaxN = 16384;  // 128 KiB of LDS for one segment

template <bool U8>
__global__ __launch_bounds__(kGeneralBlock) void stft_general(const GeneralParams p) {
    using raw_t = typename std::conditional<U8, iq_u8, cf>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    cf *const x = reinterpret_cast<cf *>(gen_smem);                       // [SPB][N]
    __shared__ double red[2 * kGeneralBlock];                              // partial sums of the segment means
    const int N = p.nperseg, LOG = p.log2n, SPB = p.segs_per_block, T = p.n_seg;
    const int tid = threadIdx.x;
    const int blocks_per_stream = (T + SPB - 1) / SPB;
    const int s = blockIdx.x / blocks_per_stream;
    const int seg0 = (blockIdx.x % blocks_per_stream) * SPB;
    if (s >= p.n_streams) return;
    const raw_t *src = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)s * p.stream_stride + (int64_t)seg0 * N;
    const int n_here = (T - seg0 < SPB) ? (T - seg0) : SPB;  // segments of this block inside the buffer
    // threads per segment: TPS = 256 / SPB (SPB divides 256: both powers of two, SPB <= 64)
    const int TPS = kGeneralBlock / SPB;
    const int q = tid / TPS, lt = tid % TPS;  // this thread's segment of the block and its place among that segment's threads
    cf *const xs = x + (int64_t)q * N;
    const bool live = q < n_here;

    // samples -> LDS at bit-reversed places, partial sums for the mean
    // (in float64: with thousands of samples under a constant offset a float32 sum strays from NumPy's pairwise one by a
    // percent of what the detrend leaves in bin 0 -- the exact sum, rounded once, stays within NumPy's own error)
    double sx = 0.0, sy = 0.0;
    if (live) {
        for (int n = lt; n < N; n += TPS) {
            const cf v = to_cf(load_iq(src + (int64_t)q * N + n));
            sx += (double)v.x;
            sy += (double)v.y;
            xs[__brev((unsigned)n) >> (32 - LOG)] = v;
        }
    }
    red[2 * tid] = sx;
    red[2 * tid + 1] = sy;
    __syncthreads();
    // the segment's mean: its TPS partial sums in a fixed order (every thread of the segment adds them the same way)
    double mxd = 0.0, myd = 0.0;
    for (int j = 0; j < TPS; ++j) {
        mxd += red[2 * (q * TPS + j)];
        myd += red[2 * (q * TPS + j) + 1];
    }
    const float mx = (float)(mxd / (double)N), my = (float)(myd / (double)N);
    // detrend='constant' (scipy _signaltools.py:3926), then the window (times sqrt(scale): the power needs no further factor)
    if (live) {
        for (int n = lt; n < N; n += TPS) {
            const int at = (int)(__brev((unsigned)n) >> (32 - LOG));
            const cf v = xs[at];
            const float w = p.window[n];
            xs[at] = cf{(v.x - mx) * w, (v.y - my) * w};
        }
    }
    __syncthreads();
    // log2 N butterfly passes
    for (int st = 1; st <= LOG; ++st) {
        const int half = 1 << (st - 1);
        const int tw_step = N >> st;  // W_(2 half)^k = W_N^(k N / (2 half))
        if (live) {
            for (int b = lt; b < N / 2; b += TPS) {
                const int k = b & (half - 1);
                const int i = ((b >> (st - 1)) << st) | k;
                const cf u = xs[i];
                const cf v = cmul(xs[i + half], p.tw[k * tw_step]);
                xs[i] = cadd(u, v);
                xs[i + half] = csub(u, v);
            }
        }
        __syncthreads();
    }
    // |X|^2 (scipy _spectral_py.py:2126-2128) -> the dense map and, for the last K segments, the look-back tail
    if (live) {
        const int seg = seg0 + q;
        float *dst = p.spec + ((int64_t)s * T + seg) * N;
        const int col = seg - (T - p.tail_cols);
        float *tdst = (p.tail && col >= 0) ? p.tail + ((int64_t)s * p.tail_cols + col) * N : nullptr;
        for (int k = lt; k < N; k += TPS) {
            const cf v = xs[k];
            const float pw = __builtin_fmaf(v.x, v.x, v.y * v.y);
            dst[k] = pw;
            if (tdst) tdst[k] = pw;
        }
    }
}

}  // namespace rt
#endif
