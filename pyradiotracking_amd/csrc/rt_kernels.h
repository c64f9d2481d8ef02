// rt_kernels.h -- gfx950 kernels of the analysis path.
//
//   stft_scan<R3, MODE>   IQ -> windowed, detrended segment FFT -> power, fused
//                         with per-bin partial row sums, the dense look-back
//                         tail and either sparse candidate emission (MODE 0),
//                         a dense spectrogram (MODE 1) or the spectrogram only
//                         (MODE 2, debug; MODE 3 = loads only, for PMC traffic
//                         calibration).  Replaces scipy.signal.spectrogram
//                         as called at radiotracking/analyze.py:234-241.
//   detect_sparse         per stream: finish row means, sort candidates,
//                         plateau extraction + statistics + shadow filter.
//   detect_dense          the same on a dense spectrogram.
//                         Both replace analyze.py:330-452 and :282-328.
//
// Segment FFT: N = 256*R3 points as radix passes 16 x 16 x R3 over a "lane
// group" of LG = N/16 lanes holding 16 points each.  With n = a + LG*m and
// k = k1 + 16*q1 + 256*q2:
//   pass 1 (in-lane over m)          A[a][k1]  = sum_m x[a+LG*m] W16^(m k1),  times W_N^(a k1)
//   exchange 1 (LDS, [k1][b][c], a = b + R3*c)
//   pass 2 (in-lane over c)          B[b][k1][q1] = sum_c A[b+R3*c][k1] W16^(c q1), times W_LG^(b q1)
//   exchange 2 (LDS, [k1][q1][b])    (R3 > 1 only)
//   pass 3 (in-lane over b)          X[k1+16*q1+256*q2] = sum_b B[b][k1][q1] W_R3^(b q2)
// For N = 256 a lane group is 16 lanes (four segments per wave64) and the whole
// transform needs one LDS exchange; no barrier is needed while LG <= 64 because
// a group then lives inside one wave.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_analyze.h"
#include "rt_core.h"
#include "rt_fft.h"

namespace rt {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int kRowF2 = 18;           // LDS exchange row stride in float2 (16 data + 2 pad = 144 B)

struct StftParams {
    const cf *iq;            // [S][stream_stride] complex64
    int64_t stream_stride;   // samples
    int32_t n_streams;
    int32_t n_seg;           // T
    int32_t segs_per_chunk;  // L
    int32_t chunks;          // ceil(T / L) chunks per stream
    int32_t blocks_per_stream;
    int32_t tail_cols;       // K
    const float *window;     // [N]
    const cf *tw1;           // [LG][16]   W_N^(a*k1)
    const cf *tw2;           // [R3][16]   W_LG^(b*q1)
    float scale;
    float thr;
    float *psum;             // [S][blocks_per_stream][N] partial row sums (one row per workgroup)
    float *tail;             // [S][K][N] trailing K columns (written)
    float *spec;             // MODE 1/2: [S][T][N]
    uint2 *hot;              // MODE 0: [S][hot_cap] (key = bin*T + t, bits of P)
    uint32_t *hot_count;     // MODE 0: [S]
    int32_t hot_cap;
};

template <int LG>
__device__ __forceinline__ void group_sync() {
    if constexpr (LG > 64) {
        __syncthreads();
    } else {
        // a lane group lives inside one wave: DS operations of a wave execute
        // in program order, only the compiler must not reorder across this.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// all-reduce (sum) over the LG lanes of a lane group
template <int LG>
__device__ __forceinline__ cf group_sum(cf v, cf *red /* [kBlock/64] LDS */) {
#pragma unroll
    for (int off = 1; off < (LG < 64 ? LG : 64); off <<= 1) {
        v.x += __shfl_xor(v.x, off, 64);
        v.y += __shfl_xor(v.y, off, 64);
    }
    if constexpr (LG > 64) {
        const int wave = threadIdx.x >> 6;
        constexpr int WPG = LG / 64;  // waves per group
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[wave] = v;
        __syncthreads();
        const int w0 = (wave / WPG) * WPG;
        cf s{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WPG; ++i) s = cadd(s, red[w0 + i]);
        v = s;
    }
    return v;
}

// bin index of result register r in lane `lt` of a group after the last pass
template <int R3>
__device__ __forceinline__ int bin_of(int lt, int r) {
    if constexpr (R3 == 1) {
        return lt + 16 * r;  // k1 = lt, q1 = r
    } else {
        constexpr int G = 16 / R3;
        const int k1 = lt / R3, qg = lt % R3;
        const int u = r / R3, q2 = r % R3;
        return k1 + 16 * (qg * G + u) + 256 * q2;
    }
}

#ifndef RT_SCAN_MIN_WAVES
#define RT_SCAN_MIN_WAVES 1
#endif

template <int R3, int MODE>
__global__ __launch_bounds__(kBlock, RT_SCAN_MIN_WAVES) void stft_scan(const StftParams p) {
    constexpr int N = 256 * R3;
    constexpr int LG = 16 * R3;
    constexpr int GPW = kBlock / LG;  // lane groups per workgroup
    constexpr int G = 16 / R3;

    __shared__ __attribute__((aligned(16))) cf xch[kBlock * kRowF2];
    __shared__ cf red[kBlock / 64];

    const int tid = threadIdx.x;
    const int g = tid / LG;
    const int lt = tid % LG;
    const int s = blockIdx.x / p.blocks_per_stream;
    const int cb = blockIdx.x % p.blocks_per_stream;
    const int chunk = cb * GPW + g;
    const bool chunk_ok = chunk < p.chunks;
    const int c0 = chunk * p.segs_per_chunk;
    const int T = p.n_seg;
    const int L = p.segs_per_chunk;

    cf *gx = xch + g * LG * kRowF2;  // this group's exchange rows

    // window and pass twiddles staged in LDS, laid out in the order the lanes
    // read them (16-byte pieces, consecutive lanes -> consecutive pieces), and
    // read just in time: keeping them in VGPRs would cost 62 registers per lane
    // and a wave per SIMD of occupancy.
    __shared__ __attribute__((aligned(16))) float4 w_lds[4 * LG];    // [m/4][lane]: w[lane + LG*(4*(m/4) + 0..3)]
    __shared__ __attribute__((aligned(16))) float4 t1_lds[8 * LG];   // [k/2][lane]: (tw1[lane][2*(k/2)], tw1[lane][2*(k/2)+1])
    __shared__ __attribute__((aligned(16))) float4 t2_lds[R3 > 1 ? 8 * R3 : 1];  // [q/2][b]
    for (int idx = tid; idx < 4 * LG; idx += kBlock) {
        const int mm = idx / LG, l = idx % LG;
        w_lds[idx] = make_float4(p.window[l + LG * (4 * mm)], p.window[l + LG * (4 * mm + 1)],
                                 p.window[l + LG * (4 * mm + 2)], p.window[l + LG * (4 * mm + 3)]);
    }
    for (int idx = tid; idx < 8 * LG; idx += kBlock) {
        const int kk = idx / LG, l = idx % LG;
        const cf a = p.tw1[l * 16 + 2 * kk], b = p.tw1[l * 16 + 2 * kk + 1];
        t1_lds[idx] = make_float4(a.x, a.y, b.x, b.y);
    }
    if constexpr (R3 > 1) {
        for (int idx = tid; idx < 8 * R3; idx += kBlock) {
            const int kk = idx / R3, b = idx % R3;
            const cf x = p.tw2[b * 16 + 2 * kk], y = p.tw2[b * 16 + 2 * kk + 1];
            t2_lds[idx] = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    __syncthreads();

    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    uint32_t next_hot = 0;  // hot bits of the segment one later in time (MODE 0)

    const cf *stream_iq = p.iq + (int64_t)s * p.stream_stride;
    const int i_first = (MODE == 0 || MODE == 3) ? 0 : 1;  // step 0 is the halo segment c0+L (sparse only)

    // software pipeline: the 16 loads of the next segment are issued before the
    // current one is transformed, so their HBM latency hides under ~700 VALU ops.
    // Loads are unconditional (segment index clamped into the stream): lanes of
    // idle groups / past-the-end steps read valid memory and discard it.
    const int seg_hi = T - 1;
    cf nxt[16];
    {
        int seg0 = c0 + L - i_first;
        seg0 = seg0 < seg_hi ? seg0 : seg_hi;
        const cf *src = stream_iq + (int64_t)seg0 * N + lt;
#pragma unroll
        for (int m = 0; m < 16; ++m) nxt[m] = src[LG * m];
    }

    for (int i = i_first; i <= L; ++i) {
        const int seg = c0 + L - i;
        const bool halo = (i == 0);
        const bool active = chunk_ok && seg < T;

        cf v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = nxt[m];
        {
            // next step's segment (the last step re-reads its own: harmless, keeps the loop uniform)
            int seg1 = (i < L) ? seg - 1 : seg;
            seg1 = seg1 < seg_hi ? seg1 : seg_hi;
            const cf *src = stream_iq + (int64_t)seg1 * N + lt;
#pragma unroll
            for (int m = 0; m < 16; ++m) nxt[m] = src[LG * m];
        }

        if constexpr (MODE == 3) {
            // traffic calibration: the scan's exact load stream, nothing else
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[0] += v[m].x + v[m].y;
            continue;
        }

        // detrend='constant': subtract the segment mean (scipy _signaltools.py:3926)
        cf sum{0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 16; ++m) sum = cadd(sum, v[m]);
        sum = group_sum<LG>(sum, red);
        const cf mean = cscale(sum, 1.0f / (float)N);
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const float4 w4 = w_lds[mm * LG + lt];
            v[4 * mm + 0] = cscale(csub(v[4 * mm + 0], mean), w4.x);
            v[4 * mm + 1] = cscale(csub(v[4 * mm + 1], mean), w4.y);
            v[4 * mm + 2] = cscale(csub(v[4 * mm + 2], mean), w4.z);
            v[4 * mm + 3] = cscale(csub(v[4 * mm + 3], mean), w4.w);
        }

        // pass 1
        dft16(v);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 t = t1_lds[kk * LG + lt];
            if (kk) v[2 * kk] = cmul(v[2 * kk], cf{t.x, t.y});
            v[2 * kk + 1] = cmul(v[2 * kk + 1], cf{t.z, t.w});
        }

        // exchange 1: element (a = lt, k1) -> row k1*R3 + b, column c
        {
            const int b = lt % R3, c = lt / R3;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) gx[(k1 * R3 + b) * kRowF2 + c] = v[k1];
        }
        group_sync<LG>();
        {
            const float4 *row = reinterpret_cast<const float4 *>(gx + lt * kRowF2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 q = row[j];
                v[2 * j] = cf{q.x, q.y};
                v[2 * j + 1] = cf{q.z, q.w};
            }
        }

        // pass 2
        dft16(v);

        if constexpr (R3 > 1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float4 t = t2_lds[kk * R3 + (lt % R3)];
                if (kk) v[2 * kk] = cmul(v[2 * kk], cf{t.x, t.y});
                v[2 * kk + 1] = cmul(v[2 * kk + 1], cf{t.z, t.w});
            }
            group_sync<LG>();  // everyone has read exchange 1
            {
                const int k1 = lt / R3, b = lt % R3;
#pragma unroll
                for (int q1 = 0; q1 < 16; ++q1)
                    gx[(k1 * R3 + q1 / G) * kRowF2 + (q1 % G) * R3 + b] = v[q1];
            }
            group_sync<LG>();
            {
                const float4 *row = reinterpret_cast<const float4 *>(gx + lt * kRowF2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 q = row[j];
                    v[2 * j] = cf{q.x, q.y};
                    v[2 * j + 1] = cf{q.z, q.w};
                }
            }
            // pass 3
            dft_groups<R3>(v);
        }
        group_sync<LG>();  // rows are free for the next segment

        // |X|^2 * scale  (scipy _spectral_py.py:2126-2128)
        float P[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) P[r] = __builtin_fmaf(v[r].x, v[r].x, v[r].y * v[r].y) * p.scale;

        if (active && !halo) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += P[r];
            if constexpr (MODE != 0) {
                float *dst = p.spec + ((int64_t)s * T + seg) * N;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[bin_of<R3>(lt, r)] = P[r];
            }
            if constexpr (MODE != 2) {
                const int col = seg - (T - p.tail_cols);
                if (col >= 0) {
                    float *dst = p.tail + ((int64_t)s * p.tail_cols + col) * N;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[bin_of<R3>(lt, r)] = P[r];
                }
            }
        }

        if constexpr (MODE == 0) {
            uint32_t hot = 0;
            if (active) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!(P[r] < p.thr)) hot |= (1u << r);
            }
            // a cell is kept if it is a candidate itself or directly precedes one (T11)
            const uint32_t emit = (active && !halo) ? (hot | next_hot) : 0u;
            if (emit) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (emit & (1u << r)) {
                        const uint32_t slot = atomicAdd(&p.hot_count[s], 1u);
                        if (slot < (uint32_t)p.hot_cap) {
                            const uint32_t key = (uint32_t)bin_of<R3>(lt, r) * (uint32_t)T + (uint32_t)seg;
                            p.hot[(int64_t)s * p.hot_cap + slot] = make_uint2(key, __float_as_uint(P[r]));
                        }
                    }
                }
            }
            next_hot = hot;
        }
    }

    if constexpr (MODE == 3) {
        if (acc[0] == 12345.678f) p.psum[0] = acc[0];  // keeps the loads alive, never true in practice
        return;
    }
    if constexpr (MODE != 2) {
        // deterministic workgroup reduction of the lane groups' row sums: one
        // partial row per workgroup (fixed summation order, no float atomics)
        __syncthreads();
        float *part = reinterpret_cast<float *>(xch);  // [GPW][N] floats = 16 KiB
#pragma unroll
        for (int r = 0; r < 16; ++r) part[g * N + bin_of<R3>(lt, r)] = chunk_ok ? acc[r] : 0.f;
        __syncthreads();
        float *dst = p.psum + ((int64_t)s * p.blocks_per_stream + cb) * N;
#pragma unroll
        for (int j = 0; j < R3; ++j) {
            const int bin = tid + kBlock * j;
            float sum = 0.f;
#pragma unroll
            for (int gg = 0; gg < GPW; ++gg) sum += part[gg * N + bin];
            dst[bin] = sum;
        }
    }
}

// ---------------------------------------------------------------------------
// detection
// ---------------------------------------------------------------------------
constexpr int kDetBlock = 1024;  // 16 waves: one stream per workgroup

struct DetectArgs {
    DetectParams dp;
    int32_t n_streams;
    int32_t n_bins;            // F
    // previous-buffer cells: prev[(s*prev_cols + (prev_cols - d))*F + f], d >= 1
    const float *prev;
    int32_t prev_cols;
    // sparse inputs
    const uint2 *hot;
    const uint32_t *hot_count;
    int32_t hot_cap;
    const float *psum;         // [S][chunks][F] partial row sums
    int32_t chunks;            // partial rows per stream
    // dense input
    const float *spec;         // [S][T][F]
    // outputs
    rt_record *records;        // pool
    int64_t pool_cap;
    int32_t rec_cap;           // per stream
    int32_t *rec_offset;       // [S]
    int32_t *rec_count;        // [S]
    unsigned long long *counters;  // [0] records allocated, [1] hot total, [2] flags
};

constexpr unsigned long long kFlagHotOverflow = 1ull;
constexpr unsigned long long kFlagRecOverflow = 2ull;
constexpr unsigned long long kFlagInconsistent = 4ull;

struct PrevCells {
    const float *base;  // points at column prev_cols of this (stream, bin): base[-d*F]
    int32_t F;
    __device__ float operator()(int32_t d) const { return base[-(int64_t)d * F]; }
};

// LDS record staging shared by both detect kernels
struct RecLds {
    rt_record *rec;       // [rec_cap]
    long long *ts_us;     // [rec_cap]
    long long *dur_us;    // [rec_cap]
    int *count;           // [1] (+ scratch words)
};

__device__ __forceinline__ RecLds carve_rec_lds(unsigned char *&ptr, int rec_cap) {
    RecLds l;
    l.ts_us = reinterpret_cast<long long *>(ptr);
    ptr += sizeof(long long) * rec_cap;
    l.dur_us = reinterpret_cast<long long *>(ptr);
    ptr += sizeof(long long) * rec_cap;
    l.rec = reinterpret_cast<rt_record *>(ptr);
    ptr += sizeof(rt_record) * rec_cap;
    l.count = reinterpret_cast<int *>(ptr);
    ptr += 16;
    return l;
}

// phase 1: a gated run becomes a record without statistics; `cell_off` tells
// phase 2 where the run's cells are (sparse: index offset into the sorted list)
__device__ __forceinline__ void push_candidate(const DetectArgs &a, RecLds &l, int s, int fi, int start, int end,
                                               float avg, int cell_off) {
    const int idx = atomicAdd(l.count, 1);
    if (idx < a.rec_cap) {
        rt_record r;
        r.stream = s;
        r.fi = fi;
        r.start = start;
        r.end = end;
        r.max_p = 0.f;
        r.mean_p = 0.f;
        r.std_db = 0.f;
        r.row_mean = avg;
        r.shadowed = 0;
        r.reserved = cell_off;
        l.rec[idx] = r;
        l.ts_us[idx] = timedelta_us(start_time(a.dp, start));
        l.dur_us[idx] = timedelta_us(run_duration(a.dp, start, end));
    }
}

// phase 2: np.max / np.mean / np.std(dB(.)) of one plateau by one wave, in the
// canonical order of rt::run_stats (64 interleaved partials, halving fold).
template <class Cell>
__device__ __forceinline__ RunStats run_stats_wave(int n, Cell cell) {
    const int lane = threadIdx.x & 63;
    double ps = 0.0, pd = 0.0;
    float pm = -INFINITY;
    int any_nan = 0;
    for (int k = lane; k < n; k += 64) {
        const float v = cell(k);
        ps += (double)v;
        pd += (double)db10(v);
        if (v != v) any_nan = 1;
        if (v > pm) pm = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ps += __shfl_xor(ps, off, 64);
        pd += __shfl_xor(pd, off, 64);
        const float o = __shfl_xor(pm, off, 64);
        if (o > pm) pm = o;
        any_nan |= __shfl_xor(any_nan, off, 64);
    }
    ps = __shfl(ps, 0, 64);
    pd = __shfl(pd, 0, 64);
    pm = __shfl(pm, 0, 64);
    const double mean_db = pd / (double)n;
    double pa = 0.0;
    for (int k = lane; k < n; k += 64) {
        const double d = (double)db10(cell(k)) - mean_db;
        pa += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pa += __shfl_xor(pa, off, 64);
    pa = __shfl(pa, 0, 64);
    RunStats r;
    r.max_p = any_nan ? NAN : pm;
    r.mean_p = (float)(ps / (double)n);
    r.std_db = (float)sqrt(pa / (double)n);
    return r;
}

// order the stream's records by (fi, start), apply the shadow filter against
// the unfiltered list (analyze.py:325) and publish them.
__device__ void publish_records(const DetectArgs &a, RecLds &l, int s, int n) {
    int *lds_base = l.count + 1;
    if (threadIdx.x == 0) {
        long long base = (long long)atomicAdd(&a.counters[0], (unsigned long long)n);
        if (base + n > a.pool_cap) {
            atomicOr(&a.counters[2], kFlagRecOverflow);
            base = -1;
        }
        *lds_base = (int)base;
        a.rec_offset[s] = base < 0 ? 0 : (int)base;
        a.rec_count[s] = base < 0 ? 0 : n;
    }
    __syncthreads();
    const int base = *lds_base;
    if (base < 0) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int rank, shadow;
        rank_and_shadow(i, n, l.rec, l.ts_us, l.dur_us, a.dp.cal_db, &rank, &shadow);
        rt_record out = l.rec[i];
        out.shadowed = shadow;
        out.reserved = 0;
        a.records[(int64_t)base + rank] = out;
    }
}

// clamp the candidate count after phase 1 (all threads), flag truncation
__device__ __forceinline__ int settled_count(const DetectArgs &a, RecLds &l) {
    __syncthreads();
    int n = *l.count;
    if (n > a.rec_cap) {
        if (threadIdx.x == 0) atomicOr(&a.counters[2], kFlagRecOverflow);
        n = a.rec_cap;
    }
    return n;
}

// One workgroup per stream.  Dynamic LDS: records | avg[F] | keys[n2] | vals[n2] | above[n2]
__global__ __launch_bounds__(kDetBlock) void detect_sparse(const DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int F = a.n_bins;
    const int T = a.dp.n_seg;

    unsigned char *ptr = smem;
    RecLds l = carve_rec_lds(ptr, a.rec_cap);
    float *avg = reinterpret_cast<float *>(ptr);
    ptr += sizeof(float) * ((F + 3) & ~3);
    uint32_t *keys = reinterpret_cast<uint32_t *>(ptr);

    const uint32_t n_raw = a.hot_count[s];
    if (tid == 0) {
        *l.count = 0;
        atomicAdd(&a.counters[1], (unsigned long long)n_raw);
    }
    if (n_raw > (uint32_t)a.hot_cap || n_raw == 0) {
        if (tid == 0) {
            if (n_raw) atomicOr(&a.counters[2], kFlagHotOverflow);
            a.rec_offset[s] = 0;
            a.rec_count[s] = 0;
        }
        return;
    }
    const int n = (int)n_raw;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    float *vals = reinterpret_cast<float *>(keys + n2);
    unsigned char *above = reinterpret_cast<unsigned char *>(vals + n2);

    // row means: np.mean(row) (analyze.py:375) from the scan's partial sums
    for (int f = tid; f < F; f += kDetBlock) {
        double sum = 0.0;
        const float *ps = a.psum + (int64_t)s * a.chunks * F + f;
        for (int c = 0; c < a.chunks; ++c) sum += (double)ps[(int64_t)c * F];
        avg[f] = (float)sum / (float)T;
    }
    for (int i = tid; i < n2; i += kDetBlock) {
        if (i < n) {
            const uint2 e = a.hot[(int64_t)s * a.hot_cap + i];
            keys[i] = e.x;
            vals[i] = __uint_as_float(e.y);
        } else {
            keys[i] = 0xFFFFFFFFu;
            vals[i] = 0.f;
        }
    }
    __syncthreads();

    // bitonic sort by key (keys are unique: one entry per cell)
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += kDetBlock) {
                // t-th compare-exchange pair of this step: i has bit j clear
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const bool up = ((i & k) == 0);
                const uint32_t ki = keys[i], kj = keys[ixj];
                if ((ki > kj) == up) {
                    keys[i] = kj;
                    keys[ixj] = ki;
                    const float tv = vals[i];
                    vals[i] = vals[ixj];
                    vals[ixj] = tv;
                }
            }
            __syncthreads();
        }
    }

    const DetectParams &dp = a.dp;
    // the predicate once per candidate cell (analyze.py:370, 378)
    for (int i = tid; i < n; i += kDetBlock) {
        const int fi = (int)(keys[i] / (uint32_t)T);
        above[i] = cell_above(vals[i], avg[fi], dp.thr, dp.snr) ? 1 : 0;
    }
    __syncthreads();

    // phase 1: maximal runs of above-cells -> gated candidates
    const int max_len = dp.tail_cols + 2;  // anything longer fails the max-duration gate
    for (int i = tid; i < n; i += kDetBlock) {
        if (!above[i]) continue;
        const uint32_t key = keys[i];
        const int fi = (int)(key / (uint32_t)T);
        const int b = (int)(key - (uint32_t)fi * (uint32_t)T);
        if (i > 0 && b > 0 && keys[i - 1] == key - 1 && above[i - 1]) continue;  // not a run start
        int j = i;
        while (j + 1 < n && (j - i) < max_len && keys[j + 1] == keys[j] + 1 && (b + (j + 1 - i)) < T && above[j + 1]) ++j;
        if ((j - i) >= max_len) continue;  // longer than any admissible signal
        const int e = b + (j - i) + 1;
        if (b > 0 && (i == 0 || keys[i - 1] != key - 1)) {
            // the cell before a run must have been emitted by the scan (T11)
            atomicOr(&a.counters[2], kFlagInconsistent);
            continue;
        }
        const float av = avg[fi];
        PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + fi, F};
        int start;
        if (gate_run(dp, b, e, av, prev, &start)) push_candidate(a, l, s, fi, start, e, av, i - b);
    }
    const int nrec = settled_count(a, l);

    // phase 2: statistics, one wave per candidate
    for (int c = tid >> 6; c < nrec; c += kDetBlock / 64) {
        rt_record &r = l.rec[c];
        const int start = r.start, off = r.reserved;
        PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + r.fi, F};
        auto cell = [&](int k) -> float {
            const int t = start + k;
            return t < 0 ? prev(-t) : vals[off + t];
        };
        const RunStats st = run_stats_wave(r.end - start, cell);
        if ((tid & 63) == 0) {
            r.max_p = st.max_p;
            r.mean_p = st.mean_p;
            r.std_db = st.std_db;
        }
    }
    __syncthreads();
    publish_records(a, l, s, nrec);
}

// One workgroup per stream.  Phase 1: one thread per bin (strided) scans its
// row sequentially in time -- the reference's row scan (analyze.py:357-450) in
// run-based form.  Phase 2/3 as in detect_sparse.
__global__ __launch_bounds__(kDetBlock) void detect_dense(const DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int F = a.n_bins;
    const int T = a.dp.n_seg;
    unsigned char *ptr = smem;
    RecLds l = carve_rec_lds(ptr, a.rec_cap);
    if (tid == 0) *l.count = 0;
    __syncthreads();

    const DetectParams &dp = a.dp;
    const float *sp = a.spec + (int64_t)s * T * F;
    for (int fi = tid; fi < F; fi += kDetBlock) {
        const float *row = sp + fi;
        PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + fi, F};
        auto cur = [&](int t) -> float { return row[(int64_t)t * F]; };
        double row_sum = -1.0;
        if (a.psum) {  // same partial sums (and bits) as the sparse path
            row_sum = 0.0;
            const float *ps = a.psum + (int64_t)s * a.chunks * F + fi;
            for (int c = 0; c < a.chunks; ++c) row_sum += (double)ps[(int64_t)c * F];
        }
        float av = 0.f;
        auto on_run = [&](int b, int e, float avg) {
            int start;
            if (gate_run(dp, b, e, avg, prev, &start)) push_candidate(a, l, s, fi, start, e, avg, 0);
        };
        scan_dense_row(dp, cur, row_sum, &av, on_run);
    }
    const int nrec = settled_count(a, l);

    for (int c = tid >> 6; c < nrec; c += kDetBlock / 64) {
        rt_record &r = l.rec[c];
        const int start = r.start;
        const float *row = sp + r.fi;
        PrevCells prev{a.prev + ((int64_t)s * a.prev_cols + a.prev_cols) * F + r.fi, F};
        auto cell = [&](int k) -> float {
            const int t = start + k;
            return t < 0 ? prev(-t) : row[(int64_t)t * F];
        };
        const RunStats st = run_stats_wave(r.end - start, cell);
        if ((tid & 63) == 0) {
            r.max_p = st.max_p;
            r.mean_p = st.mean_p;
            r.std_db = st.std_db;
        }
    }
    __syncthreads();
    publish_records(a, l, s, nrec);
}

}  // namespace rt
#endif
