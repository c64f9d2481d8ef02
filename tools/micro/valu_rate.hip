// Microbenchmark: VALU issue rate of wave64 float32 add/sub/fma per SIMD at 1..4 waves per SIMD (gfx950).
// Answers: is a wave64 v_add_f32 / v_fma_f32 a 2-cycle or a 4-cycle instruction for a SIMD that has several
// waves to pick from?  (tools only; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS, int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, long long *cyc) {
    float v[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) v[i] = threadIdx.x * 0.001f + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (KIND == 0) v[i] = __builtin_fmaf(v[i], a, b);
                else if (KIND == 1) v[i] = v[i] + a;
                else { v[i] = v[i] * a; }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int KIND>
void run(const char *name, int wps) {
    int blocks = 256 * wps;  // 256 CUs x wps workgroups of 4 waves -> wps waves per SIMD
    float *out; long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, KIND><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f, cyc);
    hipEventRecord(e0);
    k<CHAINS, KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long *h = (long long *)malloc(blocks * 8); hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
    double inst_per_wave = (double)iters * 8 * CHAINS;
    // per SIMD: wps waves each issue inst_per_wave instructions in `mean` cycles
    printf("%-6s chains %2d waves/SIMD %d: %.2f cycles per instruction per SIMD (wave view %.2f), %.3f ms\n", name, CHAINS, wps,
           mean / (inst_per_wave * wps), mean / inst_per_wave, ms);
    hipFree(out); hipFree(cyc); free(h);
}

int main() {
    for (int wps = 1; wps <= 4; ++wps) {
        run<8, 0>("fma", wps);
        run<8, 1>("add", wps);
        run<8, 2>("mul", wps);
        run<1, 0>("fma", wps);
        run<2, 0>("fma", wps);
    }
    return 0;
}
