// Microbenchmark: a segment of PPL x PPL points transformed in TWO radix-PPL passes by PPL lanes with PPL points each --
// 4096 points by ONE wave (64 points per lane; no workgroup barrier, no register prefetch: 2 waves per SIMD cover each
// other's HBM latency), 1024 points by half a wave (32 points per lane, 3 waves per SIMD) -- one wave-private exchange
// (real and imaginary halves through the same rows).  Against the product's steps: 16 points per lane, radix
// 16 x 16 x R3, two exchanges, at nperseg 4096 four waves per segment and two workgroup barriers, register double
// buffer, 3 waves per SIMD.  Same work per segment as the sparse scan's hot path:
// load, window, transform, power, per-bin row sums, threshold test.  Question: does the shape without barriers stream
// more than the product's 4.5 TB/s at config 5 (1024 streams x 781 segments) / 5.6 TB/s at config 3?  (tools only; not part of the product)
// Build and run on the GPU box:
//   hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Ipyradiotracking_amd/csrc -o /tmp/fft4096_wave tools/micro/fft4096_wave.hip && /tmp/fft4096_wave
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rt_fft.h"

using namespace rt;
typedef float f4 __attribute__((ext_vector_type(4)));

// 64-point DFT in place, natural order in and out: n = n0 + 4 n', k = k' + 16 k0
__device__ __forceinline__ void micro_dft64(cf (&v)[64]) {
    // four 16-point DFTs over n' (stride 4)
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) {
        cf a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = v[n0 + 4 * j];
        dft16(a);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[n0 + 4 * j] = a[j];  // A[n0][k' = j]
    }
    // twiddles W64^(n0 k') and 4-point DFTs over n0
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
        cf a0 = v[0 + 4 * kp], a1 = v[1 + 4 * kp], a2 = v[2 + 4 * kp], a3 = v[3 + 4 * kp];
        if (kp) {
            const float c1 = (float)__builtin_cos(-2.0 * M_PI * kp / 64.0), s1 = (float)__builtin_sin(-2.0 * M_PI * kp / 64.0);
            const float c2 = (float)__builtin_cos(-2.0 * M_PI * 2 * kp / 64.0), s2 = (float)__builtin_sin(-2.0 * M_PI * 2 * kp / 64.0);
            const float c3 = (float)__builtin_cos(-2.0 * M_PI * 3 * kp / 64.0), s3 = (float)__builtin_sin(-2.0 * M_PI * 3 * kp / 64.0);
            a1 = cmul_const(a1, c1, s1);
            a2 = cmul_const(a2, c2, s2);
            a3 = cmul_const(a3, c3, s3);
        }
        dft4(a0, a1, a2, a3);
        v[0 + 4 * kp] = a0; v[1 + 4 * kp] = a1; v[2 + 4 * kp] = a2; v[3 + 4 * kp] = a3;  // X[kp + 16 k0] in slot k0 + 4 kp
    }
    // natural order: out[kp + 16 k0] <- slot [k0 + 4 kp]
    cf t[64];
#pragma unroll
    for (int kp = 0; kp < 16; ++kp)
#pragma unroll
        for (int k0 = 0; k0 < 4; ++k0) t[kp + 16 * k0] = v[k0 + 4 * kp];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = t[i];
}

// 32-point DFT in place, natural order in and out: n = n0 + 2 n', k = k' + 16 k0
__device__ __forceinline__ void dft32(cf (&v)[32]) {
#pragma unroll
    for (int n0 = 0; n0 < 2; ++n0) {
        cf a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = v[n0 + 2 * j];
        dft16(a);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[n0 + 2 * j] = a[j];
    }
    cf t[32];
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
        cf a0 = v[2 * kp], a1 = v[2 * kp + 1];
        if (kp) a1 = cmul_const(a1, (float)__builtin_cos(-2.0 * M_PI * kp / 32.0), (float)__builtin_sin(-2.0 * M_PI * kp / 32.0));
        t[kp] = cadd(a0, a1);
        t[kp + 16] = csub(a0, a1);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = t[i];
}
template <int PPL> __device__ __forceinline__ void dft_lane(cf (&v)[PPL]) {
    if constexpr (PPL == 64) micro_dft64(v); else dft32(v);
}

// N = PPL x PPL points per segment, PPL lanes per segment (a wave holds 64 / PPL segments), PPL points per lane
template <int PPL, int WPS /* waves per SIMD the launch is built for */>
__global__ __launch_bounds__(256, WPS) void ksq(const cf *iq, const float *window_t /* [lane][PPL] */, const cf *tw_a /* [16][PPL]: W^(C ka d) */,
                                                const cf *tw_b /* [PPL][C]: W^(ka c) */, float *psum, unsigned *hits, int steps_per_wave, long n_seg_total, float thr) {
    constexpr int N = PPL * PPL, SPW = 64 / PPL /* segments per wave and step */, C = PPL / 16 /* n1 = c + C d */;
    constexpr int kRow = PPL + 4;                    // exchange row stride in floats: 16-byte aligned rows, conflict-free columns
    constexpr int kSegRows = PPL * kRow + (SPW > 1 ? 32 : 0);  // (the second segment's rows start 32 banks off the first's)
    __shared__ __attribute__((aligned(16))) float xch[4][SPW * kSegRows];
    __shared__ __attribute__((aligned(16))) cf ta[16 * PPL];
    constexpr bool WIN_LDS = (PPL == 32);  // (at 4096 points the table needs the 8-wave workgroup of k4096w)
    __shared__ __attribute__((aligned(16))) float win[WIN_LDS ? N : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane % PPL, sw = lane / PPL;
    for (int i = threadIdx.x; i < 16 * PPL; i += 256) ta[i] = tw_a[i];
    if constexpr (WIN_LDS)
        for (int i = threadIdx.x; i < N; i += 256) win[i] = window_t[i];
    __syncthreads();
    cf tb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) tb[c] = tw_b[l * C + c];
    float *const rows = xch[wave] + sw * kSegRows;
    const long wave_id = (long)blockIdx.x * 4 + wave;
    float acc[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) acc[i] = 0.f;
    unsigned n_hot = 0;
    const long seg0 = wave_id * steps_per_wave * SPW + sw * steps_per_wave;  // each half-wave walks its own run of segments
    for (int it = 0; it < steps_per_wave; ++it) {
        long seg = seg0 + it;
        if (seg >= n_seg_total) seg = n_seg_total - 1;  // (wave-uniform control flow: the last wave re-reads the last segment)
        const cf *src = iq + seg * N + l;
        cf v[PPL];
        if constexpr (WIN_LDS) {
            // loads in the order the first pass consumes them (m = n0 + 2 j), window from LDS: the first 16-point transform
            // runs while the second half of the segment is still arriving
#pragma unroll
            for (int n0 = 0; n0 < 2; ++n0)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const int m = n0 + 2 * j;
                    const f2 q = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(src + PPL * m));
                    v[m] = cf{q.x, q.y};
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n0 = 0; n0 < 2; ++n0)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[n0 + 2 * j] = cscale(v[n0 + 2 * j], win[l * PPL + n0 + 2 * j]);
        } else {
#pragma unroll
            for (int m = 0; m < PPL; ++m) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 q = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(src + PPL * m));
                v[m] = cf{q.x, q.y};
            }
            const f4 *wt = reinterpret_cast<const f4 *>(window_t + l * PPL);
#pragma unroll
            for (int q = 0; q < PPL / 4; ++q) {
                const f4 w4 = wt[q];
                v[4 * q] = cscale(v[4 * q], w4.x); v[4 * q + 1] = cscale(v[4 * q + 1], w4.y);
                v[4 * q + 2] = cscale(v[4 * q + 2], w4.z); v[4 * q + 3] = cscale(v[4 * q + 3], w4.w);
            }
        }
        dft_lane<PPL>(v);  // over m: lane n1 = l now holds A[n1][ka], ka = register index
        // exchange inside the segment's PPL lanes: lane ka gets A[n1][ka] for all n1 -- real parts, then imaginary parts
        float re[PPL];
#pragma unroll
        for (int ka = 0; ka < PPL; ++ka) rows[ka * kRow + l] = v[ka].x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < PPL / 4; ++q) {
            const f4 r4 = reinterpret_cast<const f4 *>(rows + l * kRow)[q];
            re[4 * q] = r4.x; re[4 * q + 1] = r4.y; re[4 * q + 2] = r4.z; re[4 * q + 3] = r4.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ka = 0; ka < PPL; ++ka) rows[ka * kRow + l] = v[ka].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < PPL / 4; ++q) {
            const f4 r4 = reinterpret_cast<const f4 *>(rows + l * kRow)[q];
            v[4 * q] = cf{re[4 * q], r4.x}; v[4 * q + 1] = cf{re[4 * q + 1], r4.y};
            v[4 * q + 2] = cf{re[4 * q + 2], r4.z}; v[4 * q + 3] = cf{re[4 * q + 3], r4.w};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        // twiddles W_N^(ka n1), n1 = c + C d: W^(C ka d) from LDS, W^(ka c) in registers
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const cf wa = ta[d * PPL + l];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (c == 0 && d == 0) continue;
                cf x = v[c + C * d];
                if (d) x = cmul(x, wa);
                if (c) x = cmul(x, tb[c]);
                v[c + C * d] = x;
            }
        }
        dft_lane<PPL>(v);  // over n1: X[ka + PPL kb] in v[kb]
        float mx = 0.f;
#pragma unroll
        for (int kb = 0; kb < PPL; ++kb) {
            const float P = __builtin_fmaf(v[kb].x, v[kb].x, v[kb].y * v[kb].y);
            acc[kb] += P;
            mx = __builtin_fmaxf(mx, P);
        }
        if (!(mx < thr)) ++n_hot;
    }
    // row sums of this half-wave's run of segments: bin = l + PPL kb
    float *dst = psum + (wave_id * SPW + sw) * N;
#pragma unroll
    for (int kb = 0; kb < PPL; ++kb) dst[l + PPL * kb] = acc[kb];
    if (n_hot) atomicAdd(hits, n_hot);
}

// Variant for 4096 points: ONE workgroup of eight waves per CU (2 per SIMD) so that the window table (16 KiB) fits LDS next to
// the eight exchange areas: the window multiply then does not queue behind the segment's own loads (vector-memory loads
// return in order), and with the loads issued in the order the first pass consumes them (m = n0 + 4 j, n0 = 0..3) the
// first 16-point transforms run while the rest of the segment is still arriving.
__global__ __launch_bounds__(512, 1) void k4096w(const cf *iq, const float *window_t /* [lane][64] */, const cf *tw_a, const cf *tw_b, float *psum,
                                                 unsigned *hits, int steps_per_wave, long n_seg_total, float thr) {
    constexpr int PPL = 64, N = 4096, C = 4, kRow = 68;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const xch = lds;                                   // [8][64 * kRow]
    cf *const ta = reinterpret_cast<cf *>(lds + 8 * 64 * kRow);  // [16][64]
    float *const win = lds + 8 * 64 * kRow + 2 * 16 * 64;     // [64 lanes][64]: lane-major, read as float4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 64; i += 512) ta[i] = tw_a[i];
    for (int i = threadIdx.x; i < 4096; i += 512) win[i] = window_t[i];
    __syncthreads();
    cf tb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) tb[c] = tw_b[lane * C + c];
    float *const rows = xch + wave * 64 * kRow;
    const long wave_id = (long)blockIdx.x * 8 + wave;
    float acc[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) acc[i] = 0.f;
    unsigned n_hot = 0;
    const long seg0 = wave_id * steps_per_wave;
    for (int it = 0; it < steps_per_wave; ++it) {
        long seg = seg0 + it;
        if (seg >= n_seg_total) seg = n_seg_total - 1;
        const cf *src = iq + seg * N + lane;
        cf v[PPL];
#pragma unroll
        for (int n0 = 0; n0 < 4; ++n0)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const int m = n0 + 4 * j;
                const f2 q = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(src + PPL * m));
                v[m] = cf{q.x, q.y};
            }
        __builtin_amdgcn_sched_barrier(0);
        const f4 *wt = reinterpret_cast<const f4 *>(win + lane * 64);
#pragma unroll
        for (int n0 = 0; n0 < 4; ++n0) {
            // window of this quarter (w[lane + 64 (n0 + 4 j)]: table order [lane][n0][j] would be nicer; scalar reads here)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = n0 + 4 * j;
                v[m] = cscale(v[m], win[lane * 64 + m]);
            }
        }
        (void)wt;
        dft_lane<PPL>(v);
        float re[PPL];
#pragma unroll
        for (int ka = 0; ka < PPL; ++ka) rows[ka * kRow + lane] = v[ka].x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < PPL / 4; ++q) {
            const f4 r4 = reinterpret_cast<const f4 *>(rows + lane * kRow)[q];
            re[4 * q] = r4.x; re[4 * q + 1] = r4.y; re[4 * q + 2] = r4.z; re[4 * q + 3] = r4.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ka = 0; ka < PPL; ++ka) rows[ka * kRow + lane] = v[ka].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < PPL / 4; ++q) {
            const f4 r4 = reinterpret_cast<const f4 *>(rows + lane * kRow)[q];
            v[4 * q] = cf{re[4 * q], r4.x}; v[4 * q + 1] = cf{re[4 * q + 1], r4.y};
            v[4 * q + 2] = cf{re[4 * q + 2], r4.z}; v[4 * q + 3] = cf{re[4 * q + 3], r4.w};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const cf wa = ta[d * PPL + lane];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (c == 0 && d == 0) continue;
                cf x = v[c + C * d];
                if (d) x = cmul(x, wa);
                if (c) x = cmul(x, tb[c]);
                v[c + C * d] = x;
            }
        }
        dft_lane<PPL>(v);
        float mx = 0.f;
#pragma unroll
        for (int kb = 0; kb < PPL; ++kb) {
            const float P = __builtin_fmaf(v[kb].x, v[kb].x, v[kb].y * v[kb].y);
            acc[kb] += P;
            mx = __builtin_fmaxf(mx, P);
        }
        if (!(mx < thr)) ++n_hot;
    }
    float *dst = psum + wave_id * N;
#pragma unroll
    for (int kb = 0; kb < PPL; ++kb) dst[lane + PPL * kb] = acc[kb];
    if (n_hot) atomicAdd(hits, n_hot);
}

static void run_w(const cf *iq, const float *wt, const cf *ta, const cf *tb, float *psum, unsigned *hits, long n_seg, int spw) {
    const size_t lds_bytes = (size_t)(8 * 64 * 68 + 2 * 16 * 64 + 4096) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k4096w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        printf("k4096w: %zu bytes of LDS refused\n", lds_bytes);
        return;
    }
    const long waves = (n_seg + spw - 1) / spw;
    const int blocks = (int)((waves + 7) / 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) k4096w<<<blocks, 512, lds_bytes>>>(iq, wt, ta, tb, psum, hits, spw, n_seg, 1e30f);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) k4096w<<<blocks, 512, lds_bytes>>>(iq, wt, ta, tb, psum, hits, spw, n_seg, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("nperseg 4096, window in LDS, loads in first-pass order, one 8-wave workgroup per CU, %3d segments per wave (%d workgroups): %.3f ms  %.2f TB/s  (%s)\n", spw, blocks, ms,
           (double)n_seg * 4096 * 8 / (ms * 1e-3) * 1e-12, hipGetErrorString(hipGetLastError()));
}

// reference: one segment on the host in double precision
static void host_dft(int N, const std::vector<float> &x, const std::vector<float> &w, std::vector<double> &P) {
    for (int k = 0; k < N; ++k) {
        double sr = 0, si = 0;
        for (int n = 0; n < N; ++n) {
            const double a = -2.0 * M_PI * (double)((long)n * k % N) / N;
            const double xr = (double)x[2 * n] * w[n], xi = (double)x[2 * n + 1] * w[n];
            sr += xr * cos(a) - xi * sin(a);
            si += xr * sin(a) + xi * cos(a);
        }
        P[k] = sr * sr + si * si;
    }
}

template <int PPL, int WPS>
static void run(const cf *iq, const float *wt, const cf *ta, const cf *tb, float *psum, unsigned *hits, long n_seg, int spw) {
    constexpr int SPW = 64 / PPL;
    const long waves = (n_seg + (long)spw * SPW - 1) / ((long)spw * SPW);
    const int blocks = (int)((waves + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) ksq<PPL, WPS><<<blocks, 256>>>(iq, wt, ta, tb, psum, hits, spw, n_seg, 1e30f);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) ksq<PPL, WPS><<<blocks, 256>>>(iq, wt, ta, tb, psum, hits, spw, n_seg, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("nperseg %4d, %d lanes per segment, %d waves per SIMD, %3d segments per lane group (%d workgroups): %.3f ms  %.2f TB/s  (%s)\n", PPL * PPL, PPL, WPS, spw,
           blocks, ms, (double)n_seg * PPL * PPL * 8 / (ms * 1e-3) * 1e-12, hipGetErrorString(hipGetLastError()));
}

template <int PPL>
static void shape(long S, long T, const char *what) {
    constexpr int N = PPL * PPL, C = PPL / 16;
    const long n_seg = S * T;
    const size_t bytes = (size_t)n_seg * N * sizeof(cf);
    printf("--- %s: %ld streams x %ld segments of %d samples = %.1f GB\n", what, S, T, N, bytes * 1e-9);
    cf *iq; float *wt, *psum; cf *ta, *tb; unsigned *hits;
    if (hipMalloc(&iq, bytes) != hipSuccess) { printf("no memory\n"); return; }
    const int rep = 8 * 4096 / N;  // distinct random segments
    std::vector<float> h((size_t)N * 2 * rep);
    srand(1);
    for (auto &x : h) x = (float)rand() / RAND_MAX - 0.5f;
    // every segment = one of a few random ones (the arithmetic does not care; HBM does not see a pattern it could cache)
    {
        const size_t piece = (size_t)rep * N * sizeof(cf);
        hipMemcpy(iq, h.data(), piece, hipMemcpyHostToDevice);
        size_t have = piece;
        while (have < bytes) { const size_t n = std::min(have, bytes - have); hipMemcpy((char *)iq + have, iq, n, hipMemcpyDeviceToDevice); have += n; }
    }
    std::vector<float> w(N), w_t(N);
    for (int n = 0; n < N; ++n) w[n] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * n / (double)N));
    for (int l = 0; l < PPL; ++l) for (int m = 0; m < PPL; ++m) w_t[l * PPL + m] = w[l + PPL * m];
    std::vector<cf> hta(16 * PPL), htb(PPL * C);
    for (int d = 0; d < 16; ++d) for (int ka = 0; ka < PPL; ++ka) { const double a = -2.0 * M_PI * ((double)C * ka * d) / N; hta[d * PPL + ka] = cf{(float)cos(a), (float)sin(a)}; }
    for (int ka = 0; ka < PPL; ++ka) for (int c = 0; c < C; ++c) { const double a = -2.0 * M_PI * ((double)ka * c) / N; htb[ka * C + c] = cf{(float)cos(a), (float)sin(a)}; }
    hipMalloc(&wt, N * 4); hipMalloc(&ta, sizeof(cf) * hta.size()); hipMalloc(&tb, sizeof(cf) * htb.size());
    hipMalloc(&psum, (size_t)(n_seg / 4 + 1024) * N * sizeof(float)); hipMalloc(&hits, 4); hipMemset(hits, 0, 4);
    hipMemcpy(wt, w_t.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(ta, hta.data(), sizeof(cf) * hta.size(), hipMemcpyHostToDevice);
    hipMemcpy(tb, htb.data(), sizeof(cf) * htb.size(), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    // correctness: one wave, one step: segment 0 in the first lane group
    ksq<PPL, 2><<<1, 256>>>(iq, wt, ta, tb, psum, hits, 1, 64 / PPL, 1e30f);
    std::vector<float> got(N);
    hipMemcpy(got.data(), psum, N * 4, hipMemcpyDeviceToHost);
    std::vector<float> x0(h.begin(), h.begin() + 2 * N);
    std::vector<double> P(N);
    host_dft(N, x0, w, P);
    double worst = 0;
    for (int k = 0; k < N; ++k) worst = std::max(worst, fabs(got[k] - P[k]) / (P[k] + 1e-3));
    printf("one segment against a float64 DFT: worst relative power difference %.2e\n", worst);
    if constexpr (PPL == 64) {
        for (int spw : {16, 32, 64}) run<PPL, 2>(iq, wt, ta, tb, psum, hits, n_seg, spw);
        run<PPL, 1>(iq, wt, ta, tb, psum, hits, n_seg, 32);
        // correctness of the variant, then its timing
        hipMemset(psum, 0, N * 4);
        {
            const size_t lds_bytes = (size_t)(8 * 64 * 68 + 2 * 16 * 64 + 4096) * sizeof(float);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k4096w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            k4096w<<<1, 512, lds_bytes>>>(iq, wt, ta, tb, psum, hits, 1, 1, 1e30f);
            hipMemcpy(got.data(), psum, N * 4, hipMemcpyDeviceToHost);
            double worst2 = 0;
            for (int k = 0; k < N; ++k) worst2 = std::max(worst2, fabs(got[k] - P[k]) / (P[k] + 1e-3));
            printf("variant, one segment against a float64 DFT: worst relative power difference %.2e (%s)\n", worst2, hipGetErrorString(hipGetLastError()));
        }
        for (int spw : {16, 32, 64}) run_w(iq, wt, ta, tb, psum, hits, n_seg, spw);
    } else {
        for (int spw : {16, 32, 64}) run<PPL, 3>(iq, wt, ta, tb, psum, hits, n_seg, spw);
        for (int spw : {32}) run<PPL, 2>(iq, wt, ta, tb, psum, hits, n_seg, spw);
        for (int spw : {32}) run<PPL, 4>(iq, wt, ta, tb, psum, hits, n_seg, spw);
    }
    hipFree(iq); hipFree(wt); hipFree(ta); hipFree(tb); hipFree(psum); hipFree(hits);
}

int main() {
    shape<64>(1024, 781, "config-5 share of one GPU");
    shape<32>(4096, 2343, "config 3");
    return 0;
}
