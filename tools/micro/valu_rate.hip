// Microbenchmark: vector-instruction issue rate of one SIMD at 1..4 waves per SIMD (gfx950).
// Answers, for the scan kernels' bound model:
//   * what a SIMD retires per cycle of independent wave64 v_fma / v_add / v_mul_f32 when it has 1, 2, 3 or 4 waves
//     to pick from (the scan kernels run at 3);
//   * whether the packed forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two float32 results per lane and
//     instruction) retire at the rate of the scalar forms -- i.e. whether packing the butterflies could halve the
//     issue slots of the transform -- or cost the slots of the two instructions they replace;
//   * the same with the scan kernels' instruction mix: 14 vector instructions to one ds_read_b128 / ds_write_b64.
// Build and run on the GPU box:  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o /tmp/valu_rate tools/micro/valu_rate.hip && /tmp/valu_rate
// (tools only; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

// KIND 0 fma, 1 add, 2 mul (scalar forms); 3 pk_fma, 4 pk_mul, 5 pk_add (inline asm: one instruction each, whatever
// the compiler would have made of a vector expression); 6 = fma stream with one LDS access per 14 instructions
template <int CHAINS, int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, long long *cyc) {
    __shared__ float4 lds[256 * 2];
    float v[CHAINS];
    f2 p[CHAINS];
    const f2 a2 = {a, a * 1.00001f}, b2 = {b, b * 0.5f};
    float av = a, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));  // operands in VGPRs (no constant-bus limit in the loop)
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
        v[i] = threadIdx.x * 0.001f + i;
        p[i] = f2{v[i], v[i] + 0.5f};
    }
    lds[threadIdx.x] = make_float4(a, b, a, b);
    lds[256 + threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if constexpr (KIND == 0 || KIND == 6) v[i] = __builtin_fmaf(v[i], av, bv);
                else if constexpr (KIND == 1) v[i] = v[i] + av;
                else if constexpr (KIND == 2) v[i] = v[i] * av;
                else if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
                else if constexpr (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
                else if constexpr (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
            }
            if constexpr (KIND == 6) {
                // the scan step's mix: ~16 ds_write_b64 + 8 ds_read_b128 per exchange against ~300 vector instructions
                if (r % 2 == 0) {
                    const float4 q = lds[(threadIdx.x + r) & 255];
                    v[0] += q.x;
                } else {
                    reinterpret_cast<float2 *>(lds)[512 + ((threadIdx.x * 2 + r) & 511)] = make_float2(v[1], v[2]);
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += v[i] + p[i].x + p[i].y;
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
    const double flops_per_inst = (KIND == 0 || KIND == 6) ? 128 : (KIND == 3) ? 256 : (KIND >= 4) ? 128 : 64;
    // per SIMD: wps waves each issue inst_per_wave instructions in `mean` cycles
    printf("%-7s chains %2d waves/SIMD %d: %.2f cycles per instruction per SIMD (one wave's view %.2f)  %6.1f TFLOP/s chip  %.3f ms\n",
           name, CHAINS, wps, mean / (inst_per_wave * wps), mean / inst_per_wave,
           inst_per_wave * wps * 1024 * flops_per_inst / (ms * 1e-3) * 1e-12, ms);
    hipFree(out); hipFree(cyc); free(h);
}

int main() {
    for (int wps = 1; wps <= 4; ++wps) {
        run<8, 0>("fma", wps);
        run<8, 1>("add", wps);
        run<8, 2>("mul", wps);
        run<8, 3>("pk_fma", wps);
        run<8, 4>("pk_mul", wps);
        run<8, 5>("pk_add", wps);
        run<14, 6>("fma+lds", wps);
        run<1, 0>("fma", wps);
        run<2, 0>("fma", wps);
        run<4, 0>("fma", wps);
    }
    return 0;
}
