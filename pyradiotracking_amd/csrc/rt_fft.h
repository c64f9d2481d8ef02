// rt_fft.h -- in-register small DFTs (forward, exp(-2*pi*i*n*k/R)) used as the
// radix passes of the segment FFT.  All indices are compile-time constants so
// every value lives in a VGPR pair; nothing here touches memory.
#ifndef RT_FFT_H
#define RT_FFT_H

#include <hip/hip_runtime.h>

namespace rt {

// Two forms of a complex value, the same IEEE operations in the same order (bit-identical spectra):
//   cf   a struct of two floats: scalar v_add / v_mul / v_fma_f32 on any two registers;
//   cfv  a two-element vector in an aligned VGPR pair: complex additions are one v_pk_add_f32 each, quarter turns and the
//        late 1/sqrt2 one v_pk_fma_f32 with a constant pair, constant twiddles two packed operations -- 16 % fewer vector
//        instructions in the nperseg-256 step, at the price of ~40 more registers (pairs).  A packed instruction holds
//        the SIMD ~1.45 x as long as a scalar one, so the step gains 1.7 %, and only where the registers are free: the
//        complex64 kernels of nperseg 256 use it (stft_scan: PK), every other instantiation keeps the scalar form
//        (spills at nperseg >= 512, the four-workgroup limit of the uint8 kernels; EXPERIMENTS.md, round 3, entry 20).
struct cf {
    float x, y;
};
typedef float cfv __attribute__((ext_vector_type(2)));
template <class C> __device__ __forceinline__ C make_c(float x, float y);
template <> __device__ __forceinline__ cf make_c<cf>(float x, float y) { return cf{x, y}; }
template <> __device__ __forceinline__ cfv make_c<cfv>(float x, float y) { return cfv{x, y}; }

// ---- scalar form
__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
// Contraction is spelled out (the library is built with -ffp-contract=off) so
// every instantiation of the scan kernel computes bit-identical spectra.
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return cf{__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)};
}
// multiply by -i  (forward-transform quarter turn): (x + iy)(-i) = y - ix
__device__ __forceinline__ cf mul_mi(cf a) { return cf{a.y, -a.x}; }
__device__ __forceinline__ cf cscale(cf a, float s) { return cf{a.x * s, a.y * s}; }
__device__ __forceinline__ cf cneg(cf a) { return cf{-a.x, -a.y}; }
__device__ __forceinline__ cf cmul_const(cf a, float wx, float wy) { return cmul(a, cf{wx, wy}); }
// b + (-i) a,  b - (-i) a,  b + c (-i) a,  b + c a
__device__ __forceinline__ cf add_mi(cf b, cf a) { return cadd(b, mul_mi(a)); }
__device__ __forceinline__ cf sub_mi(cf b, cf a) { return csub(b, mul_mi(a)); }
template <int SIGN>
__device__ __forceinline__ cf fma_mi(cf b, cf a, float c) {
    const cf r = mul_mi(a);
    return cf{__builtin_fmaf(r.x, SIGN * c, b.x), __builtin_fmaf(r.y, SIGN * c, b.y)};
}
__device__ __forceinline__ cf fma_s(cf b, cf a, float c) { return cf{__builtin_fmaf(a.x, c, b.x), __builtin_fmaf(a.y, c, b.y)}; }

// ---- packed form
__device__ __forceinline__ cfv cadd(cfv a, cfv b) { return a + b; }
__device__ __forceinline__ cfv csub(cfv a, cfv b) { return a - b; }
__device__ __forceinline__ cfv cmul(cfv a, cfv b) {  // (a run-time factor: its partner pair would cost two more operations than it saves)
    return cfv{__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ cfv cscale(cfv a, float s) { return a * s; }
__device__ __forceinline__ cfv cneg(cfv a) { return -a; }
__device__ __forceinline__ cfv mul_mi(cfv a) { return cfv{a.y, -a.x}; }
// a * w for a compile-time w: the partner pair (-w.y, w.x) is a constant too, so the product is two packed operations
// (a.y * -w.y == -(a.y * w.y) exactly: the same roundings as cmul)
__device__ __forceinline__ cfv cmul_const(cfv a, float wx, float wy) {
    const cfv t = a.yy * cfv{-wy, wx};
    return __builtin_elementwise_fma(a.xx, (cfv{wx, wy}), t);
}
// b + c (-i) a without forming (-i) a:  (b.x + c a.y, b.y - c a.x) -- one packed fused multiply-add on the swapped pair
// (c = +-1: the plain sums, exactly)
template <int SIGN>
__device__ __forceinline__ cfv fma_mi(cfv b, cfv a, float c) { return __builtin_elementwise_fma(a.yx, (cfv{SIGN * c, -SIGN * c}), b); }
__device__ __forceinline__ cfv add_mi(cfv b, cfv a) { return fma_mi<1>(b, a, 1.f); }
__device__ __forceinline__ cfv sub_mi(cfv b, cfv a) { return fma_mi<-1>(b, a, 1.f); }
__device__ __forceinline__ cfv fma_s(cfv b, cfv a, float c) { return __builtin_elementwise_fma(a, (cfv{c, c}), b); }

// 4-point DFT in place, natural order out.
template <class C>
__device__ __forceinline__ void dft4(C &a0, C &a1, C &a2, C &a3) {
    C s02 = cadd(a0, a2), d02 = csub(a0, a2);
    C s13 = cadd(a1, a3), e13 = csub(a1, a3);
    a0 = cadd(s02, s13);
    a2 = csub(s02, s13);
    a1 = add_mi(d02, e13);
    a3 = sub_mi(d02, e13);
}

template <class C>
__device__ __forceinline__ void dft2(C &a0, C &a1) {
    C s = cadd(a0, a1), d = csub(a0, a1);
    a0 = s;
    a1 = d;
}

#define RT_SQRT1_2 0.70710678118654752440f
#define RT_COS_PI_8 0.92387953251128675613f
#define RT_SIN_PI_8 0.38268343236508977173f

// multiply by W8^1 = (1 - i)/sqrt2 and W8^3 = (-1 - i)/sqrt2
// (a + (-i)a) = (x + y, y - x);  ((-i)a - a) = (y - x, -(x + y)) -- the same sums, then one scale
// rotations by W8^1 and W8^3 without their factor 1/sqrt2
template <class C> __device__ __forceinline__ C rot_w8_1(C a) { return add_mi(a, a); }          // (x + y, y - x)
template <class C> __device__ __forceinline__ C rot_w8_3(C a) { return add_mi(cneg(a), a); }    // (y - x, -(x + y))
template <class C> __device__ __forceinline__ C mul_w8_1(C a) { return cscale(rot_w8_1(a), RT_SQRT1_2); }
template <class C> __device__ __forceinline__ C mul_w8_3(C a) { return cscale(rot_w8_3(a), RT_SQRT1_2); }

// 8-point DFT, natural order in and out:  n = n0 + 2*n1, k = ka + 4*kb
// (4-point DFTs over n1 for each n0, twiddle W8^(n0*ka), 2-point over n0).
template <class C>
__device__ __forceinline__ void dft8(C (&v)[8]) {
    dft4(v[0], v[2], v[4], v[6]);  // n0 = 0 : Z0[ka] in v[0], v[2], v[4], v[6]
    dft4(v[1], v[3], v[5], v[7]);  // n0 = 1 : Z1[ka]
    v[3] = mul_w8_1(v[3]);         // ka = 1
    v[7] = mul_w8_3(v[7]);         // ka = 3  (ka = 2: W8^2 = -i, folded into y2 / y6)
    // Y[ka + 4*kb] = Z0[ka] + (-1)^kb Z1[ka]
    C y0 = cadd(v[0], v[1]), y4 = csub(v[0], v[1]);
    C y1 = cadd(v[2], v[3]), y5 = csub(v[2], v[3]);
    C y2 = add_mi(v[4], v[5]), y6 = sub_mi(v[4], v[5]);
    C y3 = cadd(v[6], v[7]), y7 = csub(v[6], v[7]);
    v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
    v[4] = y4; v[5] = y5; v[6] = y6; v[7] = y7;
}

// 4-point DFT whose inputs a1 and a3 still lack a factor 1/sqrt2 (the W8 rotations of the 16-point
// transform): the factor rides on the fused multiply-adds of the last butterfly level instead of
// costing multiplications of its own.
// Both odd inputs carry the factor: (a1 +- a3) unscaled, factor applied in the final level.
// ROT2: the even input a2 still lacks its quarter turn (-i), which rides on the first level's sums.
template <bool ROT2, class C>
__device__ __forceinline__ void dft4_late_odd(C &a0, C &a1, C &a2, C &a3) {
    constexpr float c = RT_SQRT1_2;
    const C s02 = ROT2 ? add_mi(a0, a2) : cadd(a0, a2), d02 = ROT2 ? sub_mi(a0, a2) : csub(a0, a2);
    const C s13 = cadd(a1, a3), e13 = csub(a1, a3);
    a0 = fma_s(s02, s13, c);
    a2 = fma_s(s02, s13, -c);
    a1 = fma_mi<1>(d02, e13, c);
    a3 = fma_mi<-1>(d02, e13, c);
}

// 4-point DFT whose input a2 still lacks the factor 1/sqrt2: s02 / d02 become fused multiply-adds.
template <class C>
__device__ __forceinline__ void dft4_late_even(C &a0, C &a1, C &a2, C &a3) {
    constexpr float c = RT_SQRT1_2;
    const C s02 = fma_s(a0, a2, c), d02 = fma_s(a0, a2, -c);
    const C s13 = cadd(a1, a3), e13 = csub(a1, a3);
    a0 = cadd(s02, s13);
    a2 = csub(s02, s13);
    a1 = add_mi(d02, e13);
    a3 = sub_mi(d02, e13);
}

// 16-point DFT, natural order in and out:  n = n0 + 4*n1, k = ka + 4*kb.
template <class C>
__device__ __forceinline__ void dft16(C (&v)[16]) {
    // 4-point DFTs over n1 for each n0: Z[n0][ka] lands in v[n0 + 4*ka]
    dft4(v[0], v[4], v[8], v[12]);
    dft4(v[1], v[5], v[9], v[13]);
    dft4(v[2], v[6], v[10], v[14]);
    dft4(v[3], v[7], v[11], v[15]);
    // twiddles W16^(n0*ka); the four that are W8 rotations keep their 1/sqrt2 for the next level
    // W16^1 = (cos pi/8, -sin pi/8), W16^3 = (sin pi/8, -cos pi/8)
    v[5] = cmul_const(v[5], RT_COS_PI_8, -RT_SIN_PI_8);     // n0=1 ka=1 : W^1
    v[9] = rot_w8_1(v[9]);                                  // n0=1 ka=2 : W^2 = W8^1   (x 1/sqrt2 late)
    v[13] = cmul_const(v[13], RT_SIN_PI_8, -RT_COS_PI_8);   // n0=1 ka=3 : W^3
    v[6] = rot_w8_1(v[6]);                                  // n0=2 ka=1 : W^2          (x 1/sqrt2 late)
    //                                                         n0=2 ka=2 : W^4 = -i     (folded into the next level)
    v[14] = rot_w8_3(v[14]);                                // n0=2 ka=3 : W^6 = W8^3   (x 1/sqrt2 late)
    v[7] = cmul_const(v[7], RT_SIN_PI_8, -RT_COS_PI_8);     // n0=3 ka=1 : W^3
    v[11] = rot_w8_3(v[11]);                                // n0=3 ka=2 : W^6          (x 1/sqrt2 late)
    v[15] = cmul_const(v[15], -RT_COS_PI_8, RT_SIN_PI_8);   // n0=3 ka=3 : W^9 = -W^1
    // 4-point DFTs over n0 for each ka: Y[ka + 4*kb] lands in v[4*ka + kb]
    dft4(v[0], v[1], v[2], v[3]);
    dft4_late_even(v[4], v[5], v[6], v[7]);         // v[6] lacks 1/sqrt2
    dft4_late_odd<true>(v[8], v[9], v[10], v[11]);  // v[9] and v[11] lack 1/sqrt2, v[10] its -i
    dft4_late_even(v[12], v[13], v[14], v[15]);     // v[14] lacks 1/sqrt2
    // transpose to natural order: out[ka + 4*kb] = v[4*ka + kb]
    C t;
    t = v[1];  v[1] = v[4];   v[4] = t;
    t = v[2];  v[2] = v[8];   v[8] = t;
    t = v[3];  v[3] = v[12];  v[12] = t;
    t = v[6];  v[6] = v[9];   v[9] = t;
    t = v[7];  v[7] = v[13];  v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// R-point DFTs over groups of R consecutive registers (R in {2,4,8,16}):
// 16/R independent transforms, natural order.
template <int R, class C>
__device__ __forceinline__ void dft_groups(C (&v)[16]) {
    if constexpr (R == 16) {
        dft16(v);
    } else if constexpr (R == 8) {
        C a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = v[i]; b[i] = v[8 + i]; }
        dft8(a);
        dft8(b);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = a[i]; v[8 + i] = b[i]; }
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
        dft4(v[4], v[5], v[6], v[7]);
        dft4(v[8], v[9], v[10], v[11]);
        dft4(v[12], v[13], v[14], v[15]);
    } else if constexpr (R == 2) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) dft2(v[i], v[i + 1]);
    }
}

// W64^k = exp(-2 pi i k / 64), k = 0 .. 47 (the exponents n0 * k' of the 64-point transform below), rounded once from double
constexpr float kW64Re[48] = {1.0f, 0.99518472f, 0.980785251f, 0.956940353f, 0.923879504f, 0.881921291f, 0.831469595f, 0.773010433f, 0.707106769f, 0.634393275f, 0.555570245f, 0.471396744f, 0.382683426f, 0.290284663f, 0.195090324f, 0.0980171412f, 0.0f, -0.0980171412f, -0.195090324f, -0.290284663f, -0.382683426f, -0.471396744f, -0.555570245f, -0.634393275f, -0.707106769f, -0.773010433f, -0.831469595f, -0.881921291f, -0.923879504f, -0.956940353f, -0.980785251f, -0.99518472f, -1.0f, -0.99518472f, -0.980785251f, -0.956940353f, -0.923879504f, -0.881921291f, -0.831469595f, -0.773010433f, -0.707106769f, -0.634393275f, -0.555570245f, -0.471396744f, -0.382683426f, -0.290284663f, -0.195090324f, -0.0980171412f};
constexpr float kW64Im[48] = {0.0f, -0.0980171412f, -0.195090324f, -0.290284663f, -0.382683426f, -0.471396744f, -0.555570245f, -0.634393275f, -0.707106769f, -0.773010433f, -0.831469595f, -0.881921291f, -0.923879504f, -0.956940353f, -0.980785251f, -0.99518472f, -1.0f, -0.99518472f, -0.980785251f, -0.956940353f, -0.923879504f, -0.881921291f, -0.831469595f, -0.773010433f, -0.707106769f, -0.634393275f, -0.555570245f, -0.471396744f, -0.382683426f, -0.290284663f, -0.195090324f, -0.0980171412f, 0.0f, 0.0980171412f, 0.195090324f, 0.290284663f, 0.382683426f, 0.471396744f, 0.555570245f, 0.634393275f, 0.707106769f, 0.773010433f, 0.831469595f, 0.881921291f, 0.923879504f, 0.956940353f, 0.980785251f, 0.99518472f};

// 64-point DFT in place, natural order in and out:  n = n0 + 4*n', k = k' + 16*k0.
// Four 16-point transforms over n' (inputs v[n0 + 4 j]: the quarter n0 of the registers -- a caller whose quarters arrive one
// after the other transforms each as it comes and calls dft64_finish), twiddles W64^(n0 k'), sixteen 4-point transforms over n0.
// dft64_finish: v[n0 + 4 k'] holds A[n0][k'], the 16-point transform of quarter n0.
template <class C>
__device__ __forceinline__ void dft64_finish(C (&v)[64]) {
    C t[64];
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
        C a0 = v[4 * kp], a1 = v[1 + 4 * kp], a2 = v[2 + 4 * kp], a3 = v[3 + 4 * kp];
        if (kp) {
            a1 = (kp == 8) ? mul_w8_1(a1) : cmul_const(a1, kW64Re[kp], kW64Im[kp]);
            a2 = (kp == 8) ? mul_mi(a2) : (kp == 4) ? mul_w8_1(a2) : (kp == 12) ? mul_w8_3(a2) : cmul_const(a2, kW64Re[2 * kp], kW64Im[2 * kp]);
            a3 = (kp == 8) ? mul_w8_3(a3) : cmul_const(a3, kW64Re[3 * kp], kW64Im[3 * kp]);
        }
        dft4(a0, a1, a2, a3);
        t[kp] = a0;  t[kp + 16] = a1;  t[kp + 32] = a2;  t[kp + 48] = a3;  // X[k' + 16 k0]
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = t[i];
}
template <class C>
__device__ __forceinline__ void dft64(C (&v)[64]) {
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) {
        C a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = v[n0 + 4 * j];
        dft16(a);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[n0 + 4 * j] = a[j];  // A[n0][k' = j]
    }
    dft64_finish(v);
}

// 32-point DFT in place, natural order in and out:  n = n0 + 2 n1, k = ka + 16 kb.
// Two 16-point transforms (even / odd inputs), W32^ka = W64^(2 ka) on the odd one's outputs, sixteen 2-point transforms.
template <class C>
__device__ __forceinline__ void dft32(C (&v)[32]) {
    C a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = v[2 * i];
        b[i] = v[2 * i + 1];
    }
    dft16(a);
    dft16(b);
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) {
        C t = b[ka];
        if (ka == 4) t = mul_w8_1(t);
        else if (ka == 8) t = mul_mi(t);
        else if (ka == 12) t = mul_w8_3(t);
        else if (ka) t = cmul_const(t, kW64Re[2 * ka], kW64Im[2 * ka]);
        v[ka] = cadd(a[ka], t);
        v[ka + 16] = csub(a[ka], t);
    }
}

}  // namespace rt
#endif
