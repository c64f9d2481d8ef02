// rt_general.h -- the spectrogram for every nperseg the fused scans do not cover: 8 and 16 in LDS (stft_general) and, by Bluestein's
// algorithm on LDS transforms of up to 16 384 points, every size that is not a power of two (stft_bluestein).
// (Round 5's stft_small -- 32 / 64 / 128 in registers -- and stft_big -- 8192 / 16 384, radix-2 in LDS -- are gone: those sizes are fused
// scans since round 6: rt_kernels.h: stft_scan<.., QS>, rt_scan_wg.h: stft_wg.)
//
// The reference hands `fft_nperseg` straight to scipy.signal.spectrogram (radiotracking/__main__.py:59,
// analyze.py:234-241): any integer.  The fused scans (rt_kernels.h, rt_scan64.h, rt_scan_wg.h) exist for the powers of two from 32 to
// 16 384; 8, 16 and every other size up to 8 192 are served here, on the dense path:
//   stft_general   x -> constant detrend -> window -> FFT -> |X|^2 * scale   (scipy _spectral_py.py:2185-2202, 2126-2128),
//                  written as the dense spectrogram [S][T][N] (+ the look-back tail of the last K segments),
// followed by detect_dense (the extractor on a dense map, any number of bins).  Two passes over 4 bytes per cell like every
// dense call: bound by its 16 bytes per sample, not a path the roofline is quoted on.
//
// Transform: radix-2 decimation in time inside LDS.  A workgroup of 256 threads holds SPB = max(1, 1024 / N) segments (at most 64);
// samples are stored at bit-reversed places, log2 N butterfly stages follow in place, two per barrier (twiddles W_N^m,
// m < N / 2, from a table made in double precision), natural order comes out.
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
// (Measured and dropped early in round 5, while the accesses at bit-reversed places still bounded these kernels: workgroups of 1 024
// threads beyond nperseg 1024 and W_(2 h)^k as the square of W_(4 h)^k instead of a second table load -- nperseg 8192 72 k -> 45 k
// MS/s, 128 150 k -> 128 k, only 16 384 gained, 50 k -> 55 k.  With those accesses gone, 512 / 1 024 threads are what stft_big and the
// long Bluestein transforms run on: rt_analyze.hip, launch_general.)

// log2 N butterfly stages of a decimation-in-time transform on `xs` (N complex values in LDS, input at bit-reversed places,
// natural order out), two stages per barrier: a thread takes the four elements base + {0, h, 2 h, 3 h} (h = the first stage's
// half-span) through both -- half the LDS traffic and barriers of a stage at a time.  An odd log2 N starts with one stage alone
// (span 2: no twiddle).  Called by all threads of the workgroup (barriers inside); `live` threads (lt of TPS per segment) work.
// U: groups a thread takes through a double stage AT ONCE -- their twiddle loads (global memory), their LDS reads, then the arithmetic
// and the writes: with one group per trip (the first version) the eight to sixteen trips of a thread at nperseg 8192 / 16 384 each waited
// for their own table loads and LDS reads, one latency after the other with two waves per SIMD to hide it.  N / 4 must be a multiple
// of TPS * U (the host picks U = min(8, N / 1024) where a workgroup holds one segment, 1 otherwise).
// U > 1 (N >= 2048): the twiddles come from two small LDS tables the caller stages once (fft_stage_tables: W_N^(64 j) and W_N^j, j < 64)
// -- W_N^m = hi[m / 64] * lo[m % 64], one more rounding of ~6e-8 where m is not a multiple of 64 (the late stages), none where it is.
// Read from the global table, every double stage was sixteen gathers per thread with a cache line per lane: the address path of the CU
// took longer than the butterflies (nperseg 8192: 72 k -> 88 k MS/s with the batching alone, -> see EXPERIMENTS.md with the tables).
constexpr int kTwSplit = 64;
__device__ __forceinline__ void fft_stage_tables(cf *hi, cf *lo, const cf *tw, int N, int tid, int nthreads) {
    for (int j = tid; j < N / 2 / kTwSplit; j += nthreads) hi[j] = tw[j * kTwSplit];
    for (int j = tid; j < kTwSplit; j += nthreads) lo[j] = tw[j];
}
// Padding: element i lives at pad_at(i, ps) = i + (i >> ps) with ps = log2 N - 6 (pad_shift) -- one spare place per N / 64 elements.  At
// bit-reversed places the 64 lanes of a wave, neighbours in the segment, lie N / 64 elements apart: unpadded, every ds_write / ds_read of
// the input scatter met in ONE bank pair, 64 cycles each; padded, neighbouring lanes land 8 bytes further on and a half-wave covers the
// 64 banks.  N + N / 2^ps + 1 places.  ps = 31: no padding (stft_general's small sizes, several segments per workgroup).
__device__ __forceinline__ int pad_at(int i, int ps) { return i + (i >> ps); }
__host__ __device__ constexpr int pad_shift(int log2n) { return log2n - 6 < 2 ? 2 : log2n - 6; }
__host__ __device__ constexpr int padded_len(int n, int log2n) { return n + (n >> pad_shift(log2n)) + 1; }
template <int U>
__device__ __forceinline__ void lds_fft_stages(cf *xs, int N, int LOG, const cf *tw, int lt, int TPS, bool live, const cf *hi = nullptr, const cf *lo = nullptr, int ps = 31) {
    auto at = [ps](int i) { return pad_at(i, ps); };
    int st = 1;
    if (LOG & 1) {
        if (live) {
            for (int b0 = lt; b0 < N / 2; b0 += TPS * U) {
                cf u[U], v[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int b = b0 + j * TPS;
                    u[j] = xs[at(2 * b)];
                    v[j] = xs[at(2 * b + 1)];
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int b = b0 + j * TPS;
                    xs[at(2 * b)] = cadd(u[j], v[j]);
                    xs[at(2 * b + 1)] = csub(u[j], v[j]);
                }
            }
        }
        __syncthreads();
        st = 2;
    }
    for (; st < LOG; st += 2) {
        const int h = 1 << (st - 1);
        const int step1 = N >> st;        // W_(2 h)^k = W_N^(k N / (2 h))
        const int step2 = N >> (st + 1);  // W_(4 h)^k = W_N^(k N / (4 h))
        if (live) {
            for (int g0 = lt; g0 < N / 4; g0 += TPS * U) {
                int i0[U];
                cf w1[U], w2[U], x0[U], x1[U], x2[U], x3[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int g = g0 + j * TPS;
                    const int k = g & (h - 1);
                    i0[j] = ((g >> (st - 1)) << (st + 1)) | k;
                    if constexpr (U == 1) {
                        w1[j] = tw[k * step1];
                        w2[j] = tw[k * step2];
                    } else {
                        const int m1 = k * step1, m2 = k * step2;
                        w1[j] = hi[m1 / kTwSplit];
                        w2[j] = hi[m2 / kTwSplit];
                        if (step1 % kTwSplit) w1[j] = cmul(w1[j], lo[m1 % kTwSplit]);  // (uniform: the late stages)
                        if (step2 % kTwSplit) w2[j] = cmul(w2[j], lo[m2 % kTwSplit]);
                    }
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    x0[j] = xs[at(i0[j])];
                    x1[j] = xs[at(i0[j] + h)];
                    x2[j] = xs[at(i0[j] + 2 * h)];
                    x3[j] = xs[at(i0[j] + 3 * h)];
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const cf a0 = x0[j], a1 = cmul(x1[j], w1[j]), a2 = x2[j], a3 = cmul(x3[j], w1[j]);
                    const cf b0 = cadd(a0, a1), b1 = csub(a0, a1), b2 = cadd(a2, a3), b3 = csub(a2, a3);  // stage st: pairs (0, h), (2 h, 3 h)
                    const cf c2 = cmul(b2, w2[j]);
                    const cf c3 = mul_mi(cmul(b3, w2[j]));  // W_(4 h)^(k + h) = -i W_(4 h)^k
                    xs[at(i0[j])] = cadd(b0, c2);               // stage st + 1: pairs (0, 2 h), (h, 3 h)
                    xs[at(i0[j] + 2 * h)] = csub(b0, c2);
                    xs[at(i0[j] + h)] = cadd(b1, c3);
                    xs[at(i0[j] + 3 * h)] = csub(b1, c3);
                }
            }
        }
        __syncthreads();
    }
}

// The same transform by decimation in FREQUENCY: natural order in, bit-reversed order out -- the mirror image of lds_fft_stages (the
// double stages in descending order, half-spans 2 h then h, twiddles behind the differences; an odd log2 N ends with the span-2 stage).
// A forward DIF transform, a pointwise product with a table kept in bit-reversed order and the DIT transform above make a circular
// convolution without a single access at bit-reversed places (stft_bluestein).
template <int U>
__device__ __forceinline__ void lds_fft_stages_dif(cf *xs, int N, int LOG, const cf *tw, int lt, int TPS, const cf *hi, const cf *lo) {
    const int first = (LOG & 1) ? 2 : 1;
    int st = first;
    while (st + 2 < LOG) st += 2;  // the DIT loop's last double stage
    for (; st >= first; st -= 2) {
        const int h = 1 << (st - 1);
        const int step1 = N >> st;        // W_(2 h)^k
        const int step2 = N >> (st + 1);  // W_(4 h)^k
        for (int g0 = lt; g0 < N / 4; g0 += TPS * U) {
            int i0[U];
            cf w1[U], w2[U], x0[U], x1[U], x2[U], x3[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int g = g0 + j * TPS;
                const int k = g & (h - 1);
                i0[j] = ((g >> (st - 1)) << (st + 1)) | k;
                if constexpr (U == 1) {
                    w1[j] = tw[k * step1];
                    w2[j] = tw[k * step2];
                } else {
                    const int m1 = k * step1, m2 = k * step2;
                    w1[j] = hi[m1 / kTwSplit];
                    w2[j] = hi[m2 / kTwSplit];
                    if (step1 % kTwSplit) w1[j] = cmul(w1[j], lo[m1 % kTwSplit]);
                    if (step2 % kTwSplit) w2[j] = cmul(w2[j], lo[m2 % kTwSplit]);
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                x0[j] = xs[i0[j]];
                x1[j] = xs[i0[j] + h];
                x2[j] = xs[i0[j] + 2 * h];
                x3[j] = xs[i0[j] + 3 * h];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                // half-span 2 h: pairs (0, 2 h), (h, 3 h); W_(4 h)^(k + h) = -i W_(4 h)^k
                const cf y0 = cadd(x0[j], x2[j]), y2 = cmul(csub(x0[j], x2[j]), w2[j]);
                const cf y1 = cadd(x1[j], x3[j]), y3 = cmul(mul_mi(csub(x1[j], x3[j])), w2[j]);
                // half-span h: pairs (0, h), (2 h, 3 h)
                xs[i0[j]] = cadd(y0, y1);
                xs[i0[j] + h] = cmul(csub(y0, y1), w1[j]);
                xs[i0[j] + 2 * h] = cadd(y2, y3);
                xs[i0[j] + 3 * h] = cmul(csub(y2, y3), w1[j]);
            }
        }
        __syncthreads();
    }
    if (LOG & 1) {
        for (int b = lt; b < N / 2; b += TPS) {
            const cf u = xs[2 * b], v = xs[2 * b + 1];
            xs[2 * b] = cadd(u, v);
            xs[2 * b + 1] = csub(u, v);
        }
        __syncthreads();
    }
}

// nperseg 8 and 16: up to 64 segments per workgroup.
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
    lds_fft_stages<1>(xs, N, LOG, p.tw, lt, TPS, live);
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

// Any OTHER nperseg (8 ... 8192, not a power of two -- the reference takes any integer, radiotracking/__main__.py:59): Bluestein's
// algorithm on the same LDS transform.  With w[n] = exp(-i pi n^2 / N),
//     X[k] = w[k] * sum_n (x[n] w[n]) conj(w[k - n]),
// a circular convolution of length M = the power of two >= 2 N - 1 with a fixed filter: A = FFT_M(x w, zero-padded),
// C = A * B (B = FFT_M of the filter, made on the host in double precision, 1 / M folded in), c = IFFT_M(C) = conj(FFT_M(conj(C))),
// X[k] = w[k] c[k] -- and since |w[k]| = 1 the power is |FFT_M(conj(C))[k]|^2 for k < N.  Detrend and window are folded into the
// first multiplication: the table `cwin` holds window[n] * sqrt(scale) * w[n].  One segment per workgroup, M complex values
// of LDS (up to 128 KiB).  Two transforms of length M >= 2 N per segment and float32 throughout: the round-off of the strongest
// bin sits 110 - 120 dB under it in every bin (SciPy's mixed-radix float32 transform: ~130 dB) -- decisions and the dB figures of
// the records are unaffected (tests: 0.01 dB), `std` over a cell that far under a tone follows the rule of DESIGN section 2 (3).
struct BluesteinParams {
    const void *iq;
    int64_t stream_stride;
    int32_t n_streams, n_seg, nperseg, m, log2m, tail_cols;
    const cf *cwin;   // [N] window * sqrt(scale) * w[n]
    const cf *bfilt;  // [M] FFT_M of the filter, divided by M, in BIT-REVERSED order (entry i = the transform's value at rev i)
    const cf *tw;     // [M / 2] W_M^j
    float *spec, *tail;
};

template <bool U8, int U = 1, int BLK = kGeneralBlock>
__global__ __launch_bounds__(BLK) void stft_bluestein(const BluesteinParams p) {
    using raw_t = typename std::conditional<U8, iq_u8, cf>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char blu_smem[];
    cf *const xs = reinterpret_cast<cf *>(blu_smem);  // [M]
    constexpr int NW = BLK / 64;
    __shared__ double red[2 * NW];
    __shared__ cf tw_hi[U > 1 ? kGeneralMaxN / 2 / kTwSplit : 1], tw_lo[U > 1 ? kTwSplit : 1];
    const int N = p.nperseg, M = p.m, LOG = p.log2m, T = p.n_seg;
    const int tid = threadIdx.x;
    if constexpr (U > 1) fft_stage_tables(tw_hi, tw_lo, p.tw, M, tid, BLK);
    const int s = blockIdx.x / T, seg = blockIdx.x % T;
    if (s >= p.n_streams) return;
    const raw_t *src = reinterpret_cast<const raw_t *>(p.iq) + (int64_t)s * p.stream_stride + (int64_t)seg * N;
    // the samples' sum (float64, a fixed order: a thread's samples, the wave's lanes by butterflies, the four waves); zero padding
    for (int j = N + tid; j < M; j += BLK) xs[j] = cf{0.f, 0.f};
    double sx = 0.0, sy = 0.0;
    for (int n = tid; n < N; n += BLK) {
        const cf v = to_cf(load_iq(src + n));
        sx += (double)v.x;
        sy += (double)v.y;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        sx += __shfl_xor(sx, d);
        sy += __shfl_xor(sy, d);
    }
    if ((tid & 63) == 0) {
        red[2 * (tid >> 6)] = sx;
        red[2 * (tid >> 6) + 1] = sy;
    }
    __syncthreads();
    double tx = 0.0, ty = 0.0;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) {
        tx += red[2 * wv];
        ty += red[2 * wv + 1];
    }
    const float mx = (float)(tx / (double)N), my = (float)(ty / (double)N);
    // (x - mean) * window * sqrt(scale) * w[n], natural order (the samples a second time, from L2)
    for (int n = tid; n < N; n += BLK) {
        const cf v = to_cf(load_iq(src + n));
        xs[n] = cmul(cf{v.x - mx, v.y - my}, p.cwin[n]);
    }
    __syncthreads();
    // No access at bit-reversed places anywhere (the first version scattered the samples there, came back for the window, and swapped
    // pairs (j, rev j) between the transforms: at those places the 64 lanes of a wave meet in one LDS bank pair -- more than half of the
    // kernel's LDS time): A by decimation in frequency, natural order in, bit-reversed out; the filter's transform is kept in that
    // order; conj(A * B) in place; decimation in time takes bit-reversed input back to natural order.
    lds_fft_stages_dif<U>(xs, M, LOG, p.tw, tid, BLK, tw_hi, tw_lo);
    for (int j = tid; j < M; j += BLK) {
        const cf c = cmul(xs[j], p.bfilt[j]);
        xs[j] = cf{c.x, -c.y};
    }
    __syncthreads();
    lds_fft_stages<U>(xs, M, LOG, p.tw, tid, BLK, true, tw_hi, tw_lo);  // FFT(conj(C)): its first N values have the spectrum's magnitudes
    float *dst = p.spec + ((int64_t)s * T + seg) * N;
    const int col = seg - (T - p.tail_cols);
    float *tdst = (p.tail && col >= 0) ? p.tail + ((int64_t)s * p.tail_cols + col) * N : nullptr;
    for (int k = tid; k < N; k += BLK) {
        const cf v = xs[k];
        const float pw = __builtin_fmaf(v.x, v.x, v.y * v.y);
        dst[k] = pw;
        if (tdst) tdst[k] = pw;
    }
}

// Row sums of the dense map the general transform wrote, one partial row per stream (DetectArgs::psum with chunks = 1): a thread per
// (stream, bin) adds its T cells in float64 (neighbouring threads read neighbouring bins: whole lines) and rounds once -- np.mean's
// float32 pairwise sum is the exact sum to a few 1e-8.  With the sums on hand detect_dense splits a row's time axis over its threads;
// without them (rt_extract: the caller's spectrogram) every thread first walks its whole row (17.6 ms against 1.5 at nperseg 128).
__global__ __launch_bounds__(256) void row_sums_dense(const float *spec, float *psum, int n_streams, int n_seg, int n_bins) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_streams * n_bins) return;
    const int s = (int)(i / n_bins), bin = (int)(i % n_bins);
    const float *row = spec + (int64_t)s * n_seg * n_bins + bin;
    double acc = 0.0;
    int t = 0;
    for (; t + 8 <= n_seg; t += 8) {  // eight loads in flight
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[(int64_t)(t + j) * n_bins];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (double)v[j];
    }
    for (; t < n_seg; ++t) acc += (double)row[(int64_t)t * n_bins];
    psum[i] = (float)acc;
}

}  // namespace rt
#endif
