// cplx.h — complex float helpers and in-register radix-4 / radix-16 butterflies
// for the gfx950 FFT kernels.  Forward transform convention everywhere:
//   X[k] = sum_n x[n] * exp(-2*pi*i*n*k/N)     (numpy.fft.fft, un-normalised;
//   reference call site app/sdr/streamer.py:119)
#pragma once
#include <hip/hip_runtime.h>

namespace sdrk {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct cf {
    float x, y;
};

__device__ __forceinline__ cf mk(float x, float y) { return cf{x, y}; }
__device__ __forceinline__ cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf operator*(cf a, float s) { return cf{a.x * s, a.y * s}; }
// a*b
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return cf{fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}
// a * (-i) = (a.y, -a.x)
__device__ __forceinline__ cf mul_mi(cf a) { return cf{a.y, -a.x}; }

// 4-point forward DFT in place: (a,b,c,d) = inputs n=0..3 -> outputs k=0..3.
__device__ __forceinline__ void bfly4(cf& a, cf& b, cf& c, cf& d) {
    cf t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2;
    c = t0 - t2;
    // k=1: t1 + (-i) t3 ; k=3: t1 + (+i) t3
    b = cf{t1.x + t3.y, t1.y - t3.x};
    d = cf{t1.x - t3.y, t1.y + t3.x};
}

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
__device__ __forceinline__ void radix16(cf (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;  // sqrt(1/2)
    // layer 1: DFT-4 over n2 for each n1; slot n1+4*k2 <- A[n1][k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) bfly4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    // internal twiddles W16^(n1*k2), n1,k2 in 1..3
    // k2 = 1: slots 5,6,7  <- W^1, W^2, W^3
    v[5] = cmul(v[5], cf{C1, -S1});
    v[6] = cf{(v[6].x + v[6].y) * R2, (v[6].y - v[6].x) * R2};  // * (1-i)/sqrt2
    v[7] = cmul(v[7], cf{S1, -C1});
    // k2 = 2: slots 9,10,11 <- W^2, W^4, W^6
    v[9] = cf{(v[9].x + v[9].y) * R2, (v[9].y - v[9].x) * R2};
    v[10] = mul_mi(v[10]);
    v[11] = cf{(v[11].y - v[11].x) * R2, -(v[11].x + v[11].y) * R2};  // * (-1-i)/sqrt2
    // k2 = 3: slots 13,14,15 <- W^3, W^6, W^9
    v[13] = cmul(v[13], cf{S1, -C1});
    v[14] = cf{(v[14].y - v[14].x) * R2, -(v[14].x + v[14].y) * R2};
    v[15] = cmul(v[15], cf{-C1, S1});
    // layer 2: DFT-4 over n1 for each k2; slot k1+4*k2 <- Y[4*k1+k2]
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) bfly4(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}

}  // namespace sdrk
