#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
// bits = bits * 16 + the four bits !(p3 < t3), !(p2 < t2), !(p1 < t1), !(p0 < t0)  (p3's the highest)
__device__ __forceinline__ unsigned shift_in4(unsigned bits, float p0, float p1, float p2, float p3, float t0, float t1, float t2, float t3) {
    unsigned long long m0, m1, m2;
    asm("v_cmp_nlt_f32_e64 %1, %7, %11\n\t"
        "v_cmp_nlt_f32_e64 %2, %6, %10\n\t"
        "v_cmp_nlt_f32_e64 %3, %5, %9\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
        "v_cmp_nlt_f32_e64 %1, %4, %8\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1"
        : "+v"(bits), "=&s"(m0), "=&s"(m1), "=&s"(m2)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(t0), "v"(t1), "v"(t2), "v"(t3)
        : "vcc");
    return bits;
}
__global__ void k(const float *p, const float *t, unsigned *out) {
    float P[16], T[16];
    for (int r = 0; r < 16; ++r) { P[r] = p[threadIdx.x * 16 + r]; T[r] = t[threadIdx.x * 16 + r]; }
    unsigned bits = 0;
#pragma unroll
    for (int q = 3; q >= 0; --q) bits = shift_in4(bits, P[4*q], P[4*q+1], P[4*q+2], P[4*q+3], T[4*q], T[4*q+1], T[4*q+2], T[4*q+3]);
    out[threadIdx.x] = bits;
}
int main() {
    std::vector<float> p(64*16), t(64*16);
    for (size_t i = 0; i < p.size(); ++i) { p[i] = (float)((i * 7919) % 13); t[i] = (float)((i * 104729) % 11); if (i % 37 == 0) p[i] = NAN; }
    float *dp, *dt; unsigned *dout;
    hipMalloc(&dp, p.size()*4); hipMalloc(&dt, t.size()*4); hipMalloc(&dout, 64*4);
    hipMemcpy(dp, p.data(), p.size()*4, hipMemcpyHostToDevice); hipMemcpy(dt, t.data(), t.size()*4, hipMemcpyHostToDevice);
    k<<<1,64>>>(dp, dt, dout);
    std::vector<unsigned> out(64); hipMemcpy(out.data(), dout, 64*4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { unsigned w = 0; for (int r = 15; r >= 0; --r) w = (w << 1) | ((p[l*16+r] < t[l*16+r]) ? 0u : 1u); if (w != out[l]) ++bad; }
    printf("bad %d\n", bad);
    return bad != 0;
}
