// Microbenchmark: a 4096-point segment transformed by ONE wave (64 points per lane, radix 64 x 64, one wave-private
// exchange, no workgroup barrier, no register prefetch: 2 waves per SIMD cover each other's HBM latency) -- against the
// product's nperseg-4096 step (four waves per segment, 16 points per lane, radix 16 x 16 x 16, two exchanges, two
// workgroup barriers, register double buffer, 3 waves per SIMD).  Same work per segment as the sparse scan's hot path:
// load, window, transform, power, per-bin row sums, threshold test.  Question: does the shape without barriers stream
// more than the product's 4.5 TB/s at config 5 (1024 streams x 781 segments)?  (tools only; not part of the product)
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
__device__ __forceinline__ void dft64(cf (&v)[64]) {
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

constexpr int kRow = 68;  // exchange row stride in floats (272 B: 16-byte aligned rows, conflict-free columns and b128 rows)

template <int WPS /* waves per SIMD the launch is built for */>
__global__ __launch_bounds__(256, WPS) void k4096(const cf *iq, const float *window_t /* [lane][64] */, const cf *tw_a /* [16][64]: W^(4 ka d) */,
                                                  const cf *tw_b /* [64][4]: W^(ka c) */, float *psum, unsigned *hits, int segs_per_wave, int n_seg_total, float thr) {
    __shared__ __attribute__((aligned(16))) float xch[4][64 * kRow];
    __shared__ __attribute__((aligned(16))) cf ta[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) ta[i] = tw_a[i];
    __syncthreads();
    cf tb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tb[c] = tw_b[lane * 4 + c];
    float *const rows = xch[wave];
    const long wave_id = (long)blockIdx.x * 4 + wave;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    unsigned n_hot = 0;
    const long seg0 = wave_id * segs_per_wave;
    for (int it = 0; it < segs_per_wave; ++it) {
        const long seg = seg0 + it;
        if (seg >= n_seg_total) break;
        const cf *src = iq + seg * 4096 + lane;
        cf v[64];
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 q = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(src + 64 * m));
            v[m] = cf{q.x, q.y};
        }
        // window (this lane's 64 coefficients side by side)
        const f4 *wt = reinterpret_cast<const f4 *>(window_t + lane * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f4 w4 = wt[q];
            v[4 * q] = cscale(v[4 * q], w4.x); v[4 * q + 1] = cscale(v[4 * q + 1], w4.y);
            v[4 * q + 2] = cscale(v[4 * q + 2], w4.z); v[4 * q + 3] = cscale(v[4 * q + 3], w4.w);
        }
        dft64(v);  // over m: lane n1 = lane now holds A[n1][ka], ka = register index
        // exchange: lane ka gets A[n1][ka] for all n1 -- real parts, then imaginary parts, through the same rows
        float re[64];
#pragma unroll
        for (int ka = 0; ka < 64; ++ka) rows[ka * kRow + lane] = v[ka].x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f4 r4 = reinterpret_cast<const f4 *>(rows + lane * kRow)[q];
            re[4 * q] = r4.x; re[4 * q + 1] = r4.y; re[4 * q + 2] = r4.z; re[4 * q + 3] = r4.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ka = 0; ka < 64; ++ka) rows[ka * kRow + lane] = v[ka].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f4 r4 = reinterpret_cast<const f4 *>(rows + lane * kRow)[q];
            v[4 * q] = cf{re[4 * q], r4.x}; v[4 * q + 1] = cf{re[4 * q + 1], r4.y};
            v[4 * q + 2] = cf{re[4 * q + 2], r4.z}; v[4 * q + 3] = cf{re[4 * q + 3], r4.w};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        // twiddles W_4096^(ka n1), n1 = c + 4 d: W^(4 ka d) from LDS, W^(ka c) in registers
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const cf wa = ta[d * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c == 0 && d == 0) continue;
                cf x = v[c + 4 * d];
                if (d) x = cmul(x, wa);
                if (c) x = cmul(x, tb[c]);
                v[c + 4 * d] = x;
            }
        }
        dft64(v);  // over n1: X[ka + 64 kb] in v[kb]
        float mx = 0.f;
#pragma unroll
        for (int kb = 0; kb < 64; ++kb) {
            const float P = __builtin_fmaf(v[kb].x, v[kb].x, v[kb].y * v[kb].y);
            acc[kb] += P;
            mx = __builtin_fmaxf(mx, P);
        }
        if (!(mx < thr)) ++n_hot;
    }
    // row sums of this wave's chunk: bin = lane + 64 kb
    float *dst = psum + wave_id * 4096;
#pragma unroll
    for (int kb = 0; kb < 64; ++kb) dst[lane + 64 * kb] = acc[kb];
    if (n_hot) atomicAdd(hits, n_hot);
}

// reference: one segment on the host in double precision
static void host_dft(const std::vector<float> &x, const std::vector<float> &w, std::vector<double> &P) {
    const int N = 4096;
    std::vector<double> re(N), im(N);
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

template <int WPS>
static void run(const cf *iq, const float *wt, const cf *ta, const cf *tb, float *psum, unsigned *hits, long n_seg, int spw) {
    const long waves = (n_seg + spw - 1) / spw;
    const int blocks = (int)((waves + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) k4096<WPS><<<blocks, 256>>>(iq, wt, ta, tb, psum, hits, spw, (int)n_seg, 1e30f);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) k4096<WPS><<<blocks, 256>>>(iq, wt, ta, tb, psum, hits, spw, (int)n_seg, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("one wave per segment, %d waves per SIMD, %3d segments per wave (%d workgroups): %.3f ms  %.2f TB/s  (%s)\n", WPS, spw, blocks, ms,
           (double)n_seg * 4096 * 8 / (ms * 1e-3) * 1e-12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const long S = 1024, T = 781, n_seg = S * T;  // the config-5 share of one GPU
    const size_t bytes = (size_t)n_seg * 4096 * sizeof(cf);
    cf *iq; float *wt, *psum; cf *ta, *tb; unsigned *hits;
    if (hipMalloc(&iq, bytes) != hipSuccess) return 1;
    std::vector<float> h((size_t)4096 * 2 * 8);
    srand(1);
    for (auto &x : h) x = (float)rand() / RAND_MAX - 0.5f;
    // every segment = one of 8 random segments (the arithmetic does not care; HBM does not see a pattern it could cache: 26 GB)
    for (long s = 0; s < n_seg; s += 8) hipMemcpyAsync(iq + s * 4096, h.data(), std::min<long>(8, n_seg - s) * 4096 * sizeof(cf), hipMemcpyHostToDevice, 0);
    std::vector<float> w(4096), w_t(4096);
    for (int n = 0; n < 4096; ++n) w[n] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * n / 4096.0));
    for (int l = 0; l < 64; ++l) for (int m = 0; m < 64; ++m) w_t[l * 64 + m] = w[l + 64 * m];
    std::vector<cf> hta(16 * 64), htb(64 * 4);
    for (int d = 0; d < 16; ++d) for (int ka = 0; ka < 64; ++ka) { const double a = -2.0 * M_PI * (4.0 * ka * d) / 4096.0; hta[d * 64 + ka] = cf{(float)cos(a), (float)sin(a)}; }
    for (int ka = 0; ka < 64; ++ka) for (int c = 0; c < 4; ++c) { const double a = -2.0 * M_PI * ((double)ka * c) / 4096.0; htb[ka * 4 + c] = cf{(float)cos(a), (float)sin(a)}; }
    hipMalloc(&wt, 4096 * 4); hipMalloc(&ta, sizeof(cf) * hta.size()); hipMalloc(&tb, sizeof(cf) * htb.size());
    hipMalloc(&psum, (size_t)(n_seg + 64) * 4096 * sizeof(float) / 8); hipMalloc(&hits, 4); hipMemset(hits, 0, 4);
    hipMemcpy(wt, w_t.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipMemcpy(ta, hta.data(), sizeof(cf) * hta.size(), hipMemcpyHostToDevice);
    hipMemcpy(tb, htb.data(), sizeof(cf) * htb.size(), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    // correctness: one wave, one segment
    k4096<2><<<1, 256>>>(iq, wt, ta, tb, psum, hits, 1, 1, 1e30f);
    std::vector<float> got(4096);
    hipMemcpy(got.data(), psum, 4096 * 4, hipMemcpyDeviceToHost);
    std::vector<float> x0(h.begin(), h.begin() + 8192);
    std::vector<double> P(4096);
    host_dft(x0, w, P);
    double worst = 0;
    for (int k = 0; k < 4096; ++k) worst = std::max(worst, fabs(got[k] - P[k]) / (P[k] + 1e-3));
    printf("one segment against a float64 DFT: worst relative power difference %.2e\n", worst);
    for (int spw : {8, 16, 32, 64}) run<2>(iq, wt, ta, tb, psum, hits, n_seg, spw);
    for (int spw : {16, 32}) run<1>(iq, wt, ta, tb, psum, hits, n_seg, spw);
    return 0;
}
