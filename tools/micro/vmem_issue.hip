// Microbenchmark: a streaming read with arithmetic between the requests, in the shape of the scan step -- per wave
// and step, request the NEXT 8 KiB (register double buffer), then V independent float32 FMAs on the CURRENT 8 KiB --
// at 3 workgroups of 4 waves per CU (LDS-limited, like the scan kernels).  Question: with the same bytes per step,
// does the width of the load instruction matter (16 x dwordx2 per lane, the scan's shape, against 8 x dwordx4), and
// how does the achieved HBM rate fall as V grows?  (tools only; not part of the product)
// Build and run on the GPU box:  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o /tmp/vmem_issue tools/micro/vmem_issue.hip && /tmp/vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// one wave reads `steps` consecutive 8 KiB pieces of its own contiguous region
template <int WIDTH /* bytes per lane and load: 8 or 16 */, int V /* FMAs per step and lane */, int SPLIT /* request in SPLIT parts spread over the step */>
__global__ __launch_bounds__(256, 3) void k(const float *src, float *out, int steps) {
    extern __shared__ float pad[];  // 48 KiB: three workgroups per CU
    constexpr int NLOAD = 8192 / 64 / WIDTH;  // loads per lane and step: 16 or 8
    constexpr int NF = 32;                    // floats per lane and step
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const char *base = reinterpret_cast<const char *>(src) + (size_t)wave * steps * 8192;
    float cur[NF], nxt[NF];
    auto request = [&](int step, int part) {
        const char *p = base + (size_t)step * 8192 + lane * WIDTH;
#pragma unroll
        for (int m = part * (NLOAD / SPLIT); m < (part + 1) * (NLOAD / SPLIT); ++m) {
            if constexpr (WIDTH == 8) {
                const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p + m * 64 * WIDTH));
                nxt[2 * m] = v.x; nxt[2 * m + 1] = v.y;
            } else {
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + m * 64 * WIDTH));
                nxt[4 * m] = v.x; nxt[4 * m + 1] = v.y; nxt[4 * m + 2] = v.z; nxt[4 * m + 3] = v.w;
            }
        }
    };
#pragma unroll
    for (int part = 0; part < SPLIT; ++part) request(0, part);
    float acc[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < NF; ++i) cur[i] = nxt[i];
        const int sn = (s + 1 < steps) ? s + 1 : s;
        constexpr int ROUNDS = V / NF;
#pragma unroll
        for (int part = 0; part < SPLIT; ++part) {
            request(sn, part);
#pragma unroll
            for (int r = part * (ROUNDS / SPLIT); r < (part + 1) * (ROUNDS / SPLIT); ++r) {
#pragma unroll
                for (int i = 0; i < NF; ++i) acc[i] = __builtin_fmaf(acc[i], 0.999f, cur[i]);
            }
            if constexpr (SPLIT > 1) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ROUNDS == 0) {
#pragma unroll
            for (int i = 0; i < NF; ++i) acc[i] += cur[i];
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) t += acc[i];
    if (t == 123.456f) out[wave * 64 + lane] = t + pad[lane];
}

// the scan's address pattern at nperseg 256: the four 16-lane groups of a wave walk four different chunks (32 segments of 2 KiB =
// 64 KiB apart), so a wave's 8 KiB per step are four 2 KiB pieces; 16 loads of 8 B per lane, each instruction = one 128-byte
// line per group
// SEQ: the four groups take four CONSECUTIVE 2 KiB segments of one run instead (a wave's 8 KiB per step are contiguous, but every
// load instruction still touches four lines 2 KiB apart) -- the pattern of an experiment with the scan kernel (EXPERIMENTS.md)
template <int V, bool SEQ = false>
__global__ __launch_bounds__(256, 3) void k_scanlike(const float *src, float *out, int steps) {
    extern __shared__ float pad[];
    constexpr int NF = 32;
    const int lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    // wave w owns four chunks: chunk c = 4 w + grp, each `steps` segments of 2 KiB
    const char *base = SEQ ? reinterpret_cast<const char *>(src) + (size_t)wave * steps * 8192 + (3 - grp) * 2048 + l16 * 8
                           : reinterpret_cast<const char *>(src) + ((size_t)wave * 4 + grp) * steps * 2048 + l16 * 8;
    const size_t step_bytes = SEQ ? 8192 : 2048;
    float cur[NF], nxt[NF];
    auto request = [&](int step) {
        const char *p = base + (size_t)step * step_bytes;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p + m * 128));
            nxt[2 * m] = v.x; nxt[2 * m + 1] = v.y;
        }
    };
    request(steps - 1);  // (the scan walks a chunk down from its latest segment)
    float acc[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = 0.f;
    for (int s = steps - 1; s >= 0; --s) {
#pragma unroll
        for (int i = 0; i < NF; ++i) cur[i] = nxt[i];
        request(s > 0 ? s - 1 : 0);
#pragma unroll
        for (int r = 0; r < V / NF; ++r)
#pragma unroll
            for (int i = 0; i < NF; ++i) acc[i] = __builtin_fmaf(acc[i], 0.999f, cur[i]);
        if constexpr (V == 0) {
#pragma unroll
            for (int i = 0; i < NF; ++i) acc[i] += cur[i];
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) t += acc[i];
    if (t == 123.456f) out[wave * 64 + lane] = t + pad[lane];
}

template <int V, bool SEQ = false>
void run_scanlike(const float *src, float *out, size_t bytes, int steps) {
    const int waves = (int)(bytes / ((size_t)steps * 8192));
    const int blocks = waves / 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_scanlike<V, SEQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_scanlike<V, SEQ><<<blocks, 256, 48 * 1024>>>(src, out, steps);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k_scanlike<V, SEQ><<<blocks, 256, 48 * 1024>>>(src, out, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double moved = (double)blocks * 4 * steps * 8192;
    if (SEQ)
        printf("four consecutive 2 KiB segments per wave and step (8 KiB contiguous, each instruction four lines 2 KiB apart), %3d FMAs per step, %2d steps per workgroup: %.3f ms  %.2f TB/s\n",
               V, steps, ms, moved / (ms * 1e-3) * 1e-12);
    else
        printf("scan-like addresses (4 x 2 KiB pieces per wave and step, %d KiB apart, walked downwards), %3d FMAs per step, %2d steps per workgroup: %.3f ms  %.2f TB/s\n",
               steps * 2, V, steps, ms, moved / (ms * 1e-3) * 1e-12);
}

template <int WIDTH, int V, int SPLIT>
void run(const float *src, float *out, size_t bytes, int steps = 64) {
    const int waves = (int)(bytes / ((size_t)steps * 8192));
    const int blocks = waves / 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<WIDTH, V, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<WIDTH, V, SPLIT><<<blocks, 256, 48 * 1024>>>(src, out, steps);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<WIDTH, V, SPLIT><<<blocks, 256, 48 * 1024>>>(src, out, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double moved = (double)blocks * 4 * steps * 8192;
    printf("load width %2d B x %2d per step, %3d FMAs per step%s, %3d steps per workgroup: %.3f ms  %.2f TB/s\n", WIDTH, 8192 / 64 / WIDTH, V,
           SPLIT > 1 ? ", requests spread over the step in 4 parts" : "", steps, ms, moved / (ms * 1e-3) * 1e-12);
}

int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB: far beyond the Infinity Cache
    float *src, *out;
    if (hipMalloc(&src, bytes) != hipSuccess) return 1;
    hipMalloc(&out, 64 << 20);
    hipMemset(src, 0x3c, bytes);
    run<8, 0, 1>(src, out, bytes);    run<16, 0, 1>(src, out, bytes);
    run<8, 256, 1>(src, out, bytes);  run<16, 256, 1>(src, out, bytes);
    run<8, 512, 1>(src, out, bytes);  run<16, 512, 1>(src, out, bytes);
    run<8, 768, 1>(src, out, bytes);  run<16, 768, 1>(src, out, bytes);
    run<8, 1024, 1>(src, out, bytes); run<16, 1024, 1>(src, out, bytes);
    run<8, 768, 4>(src, out, bytes);  run<16, 768, 4>(src, out, bytes);
    run<8, 1024, 4>(src, out, bytes); run<16, 1024, 4>(src, out, bytes);
    run<8, 768, 1>(src, out, bytes, 32);    run<8, 768, 1>(src, out, bytes, 128);  // contiguous, the workgroup lives of the rows below
    run_scanlike<0>(src, out, bytes, 32);   run_scanlike<768>(src, out, bytes, 32);
    run_scanlike<0, true>(src, out, bytes, 32);   run_scanlike<768, true>(src, out, bytes, 32);
    run_scanlike<0>(src, out, bytes, 64);   run_scanlike<768>(src, out, bytes, 64);
    run_scanlike<0>(src, out, bytes, 128);  run_scanlike<768>(src, out, bytes, 128);
    hipFree(src); hipFree(out);
    return 0;
}
