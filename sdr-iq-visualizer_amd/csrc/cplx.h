// cplx.h — complex float helpers and in-register radix-4 / radix-16 butterflies
// for the gfx950 FFT kernels.  Forward transform convention everywhere:
//   X[k] = sum_n x[n] * exp(-2*pi*i*n*k/N)     (numpy.fft.fft, un-normalised;
//   reference call site app/sdr/streamer.py:119)
#pragma once
#include <hip/hip_runtime.h>

namespace sdrk {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef SDRK_PACKED_CF
#define SDRK_PACKED_CF 1
#endif
#if SDRK_PACKED_CF
// A complex value is a two-float vector in an aligned register pair.  Sums and differences are one v_pk_add_f32, a product is
// two packed instructions, and the half swap / sign flip of a multiplication by -+i rides on the op_sel / neg modifiers of the
// instruction that consumes it (op_sel[i]: which half of source i feeds the LOW result, op_sel_hi[i]: the HIGH result).  hipcc
// folds whole-vector negation and broadcasts into those modifiers by itself but not a one-lane negation (it emits v_xor + v_mov),
// hence the few asm statements; they are plain VALU arithmetic (no lane crossing, interlocked by the hardware) and not volatile,
// so the scheduler moves them like any other instruction.  Every form below performs the SAME float operations in the same
// order as the scalar forms of the #else branch: the two builds are bit-identical (tools/hash_outputs.py under both).
// What it buys: one wave issues a VALU instruction every four cycles, packed or not, and a SIMD takes two plain float32
// instructions of two different waves in those four cycles but one packed one (tools/pkprobe.hip, profiles/r05/pkprobe.log):
// at three waves per SIMD a complex add costs 1.84 ns packed against 2 x 1.15 ns, and a wave's own dependent chain halves.
typedef v2f cf;
__device__ __forceinline__ cf mk(float x, float y) { return cf{x, y}; }
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf add_mi(cf a, cf b) {
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf add_pi(cf a, cf b) {
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a*b = (a.x b.x - [a.y b.y], a.x b.y + [a.y b.x]), the bracketed products rounded first (as the scalar form does)
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
// the same with a wave-uniform factor held in a scalar register pair (one constant-bus operand per instruction)
__device__ __forceinline__ cf cmul_k(cf a, float bx, float by) {
    const cf b = cf{bx, by};
    cf m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(m));
    return r;
}
// s * k + a, one rounding per component (an explicit fused multiply-add: the build runs with -ffp-contract=on)
__device__ __forceinline__ cf fma_s(cf s, float k, cf a) { return __builtin_elementwise_fma(s, cf{k, k}, a); }
#else
struct cf {
    float x, y;
};

__device__ __forceinline__ cf mk(float x, float y) { return cf{x, y}; }
__device__ __forceinline__ cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf operator*(cf a, float s) { return cf{a.x * s, a.y * s}; }
__device__ __forceinline__ cf add_mi(cf a, cf b) { return cf{a.x + b.y, a.y - b.x}; }
__device__ __forceinline__ cf add_pi(cf a, cf b) { return cf{a.x - b.y, a.y + b.x}; }
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return cf{fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ cf cmul_k(cf a, float bx, float by) { return cmul(a, cf{bx, by}); }
__device__ __forceinline__ cf fma_s(cf s, float k, cf a) { return cf{fmaf(s.x, k, a.x), fmaf(s.y, k, a.y)}; }
#endif

// a * (-i) = (a.y, -a.x)
__device__ __forceinline__ cf mul_mi(cf a) { return add_mi(cf{0.f, 0.f}, a); }
// (1 -+ i) a: a (1 - i) = (a.x + a.y, a.y - a.x) ; a (1 + i) = (a.x - a.y, a.y + a.x)
__device__ __forceinline__ cf one_mi(cf a) { return add_mi(a, a); }
__device__ __forceinline__ cf one_pi(cf a) { return add_pi(a, a); }
// a * (1 - i) / sqrt2 ; a * (-1 - i) / sqrt2
__device__ __forceinline__ cf rot_m45(cf a) { return one_mi(a) * 0.70710678118654752440f; }
__device__ __forceinline__ cf rot_m135(cf a) { return one_pi(a) * -0.70710678118654752440f; }

// the two sums of a radix-4 butterfly, t0 = a + c and t1 = a - c, with c = sc * k still unscaled: two fused multiply-adds
__device__ __forceinline__ void sums_scaled(cf a, cf sc, float k, cf& t0, cf& t1) {
    t0 = fma_s(sc, k, a);
    t1 = fma_s(sc, -k, a);
}
// the rest of a radix-4 butterfly from its four partial sums t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d
__device__ __forceinline__ void bfly4_finish(cf t0, cf t1, cf t2, cf t3, cf& a, cf& b, cf& c, cf& d) {
    a = t0 + t2;
    c = t0 - t2;
    b = add_mi(t1, t3);   // k=1: t1 + (-i) t3
    d = add_pi(t1, t3);   // k=3: t1 + (+i) t3
}
// 4-point forward DFT in place: (a,b,c,d) = inputs n=0..3 -> outputs k=0..3.
__device__ __forceinline__ void bfly4(cf& a, cf& b, cf& c, cf& d) { bfly4_finish(a + c, a - c, b + d, b - d, a, b, c, d); }

// 2-point DFT in place.
__device__ __forceinline__ void bfly2(cf& a, cf& b) {
    cf t = a - b;
    a = a + b;
    b = t;
}

// Register slot that holds output k of radix16() (base-4 digit reversal).
__host__ __device__ constexpr int rev16(int k) { return (k >> 2) + 4 * (k & 3); }

// 16-point forward DFT on v[0..15] (v[n] = input n).  On return output k is in
// v[rev16(k)].  Two radix-4 layers: n = n1 + 4*n2, k = 4*k1 + k2,
//   W16^(nk) = W4^(n2 k2) * W16^(n1 k2) * W4^(n1 k1).
// With WIN the inputs are x[n] still to be multiplied by the real window coefficients w[n]: half of those products ride on the
// first sums of layer 1 (x_a w_a + x_c w_c = fma(x_c, w_c, x_a w_a)).
template <bool WIN = false>
__device__ __forceinline__ void radix16(cf (&v)[16], const float* __restrict__ w = nullptr) {
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    // layer 1: DFT-4 over n2 for each n1; slot n1+4*k2 <- A[n1][k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
        if (WIN) {
            cf s0, s1, s2, s3;
            sums_scaled(v[n1] * w[n1], v[n1 + 8], w[n1 + 8], s0, s1);
            sums_scaled(v[n1 + 4] * w[n1 + 4], v[n1 + 12], w[n1 + 12], s2, s3);
            bfly4_finish(s0, s1, s2, s3, v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
        } else {
            bfly4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
        }
    }
    // internal twiddles W16^(n1*k2), n1,k2 in 1..3, and layer 2: DFT-4 over n1 for each k2; slot k1+4*k2 <- Y[4*k1+k2].
    // The factors (1 -+ i)/sqrt2 and -i are not applied by themselves: the sqrt(1/2) rides on the fused multiply-adds of the
    // butterfly's first sums and the quarter turns on the operand modifiers of its additions.
    constexpr float R2 = 0.70710678118654752440f;  // sqrt(1/2)
    cf t0, t1, t2, t3;
    // k2 = 0: no twiddles
    bfly4(v[0], v[1], v[2], v[3]);
    // k2 = 1: slots 5,6,7 <- W^1, W^2 = (1-i)/sqrt2, W^3
    v[5] = cmul_k(v[5], C1, -S1);
    v[7] = cmul_k(v[7], S1, -C1);
    sums_scaled(v[4], one_mi(v[6]), R2, t0, t1);
    t2 = v[5] + v[7];
    t3 = v[5] - v[7];
    bfly4_finish(t0, t1, t2, t3, v[4], v[5], v[6], v[7]);
    // k2 = 2: slots 9,10,11 <- W^2, W^4 = -i, W^6 = (-1-i)/sqrt2
    t0 = add_mi(v[8], v[10]);
    t1 = add_pi(v[8], v[10]);
    {
        const cf p9 = one_mi(v[9]) * R2, s11 = one_pi(v[11]);
        t2 = fma_s(s11, -R2, p9);
        t3 = fma_s(s11, R2, p9);
    }
    bfly4_finish(t0, t1, t2, t3, v[8], v[9], v[10], v[11]);
    // k2 = 3: slots 13,14,15 <- W^3, W^6 = (-1-i)/sqrt2, W^9
    v[13] = cmul_k(v[13], S1, -C1);
    v[15] = cmul_k(v[15], -C1, S1);
    sums_scaled(v[12], one_pi(v[14]), -R2, t0, t1);
    t2 = v[13] + v[15];
    t3 = v[13] - v[15];
    bfly4_finish(t0, t1, t2, t3, v[12], v[13], v[14], v[15]);
}

}  // namespace sdrk
