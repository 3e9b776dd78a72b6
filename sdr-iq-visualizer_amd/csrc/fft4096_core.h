// fft4096_core.h — the in-LDS 4096-point transform shared by the flagship kernel
// (fft4096.hip) and the strided first step of the large-N plans (fft_large.hip).
// See fft4096.hip for the decomposition and the LDS layouts.
#pragma once
#include "cplx.h"
#include "kernels.h"

namespace sdrk {

constexpr int F4K_N = 4096;
constexpr int F4K_THREADS = 256;
#ifndef F4K_WAVES
#define F4K_WAVES 3  // waves per SIMD = workgroups per CU (<=168 VGPRs)
#endif
constexpr int F4K_XCH_ELEMS = 4112;  // exchange buffer (max index 4110), 32,896 B
constexpr int F4K_TW_ELEMS = 512;    // W256^(n k) as [k][n] and W4096^tid, 2 KiB each

typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}


// Per-thread LDS addressing for the two exchanges (complex64 units).
struct F4kAddr {
    int lo, hi;
    int x1w_even, x1w_odd;  // exchange-1 write: (tid ^ 16*(k&1)) + 256 k
    int x1r_even, x1r_odd;  // exchange-1 read : lo + 16 (j ^ (hi&1)) + 256 hi
    int x2w;                // exchange-2 write: hi + 16 k + 257 lo ; read: tid + 257 j
};

__device__ __forceinline__ F4kAddr f4k_addr(int tid) {
    F4kAddr a;
    a.lo = tid & 15;
    a.hi = tid >> 4;
    a.x1w_even = tid;
    a.x1w_odd = tid ^ 16;
    const int b = a.hi & 1;
    a.x1r_even = a.lo + 256 * a.hi + 16 * b;
    a.x1r_odd = a.lo + 256 * a.hi - 16 * b;
    a.x2w = a.hi + 257 * a.lo;
    return a;
}

// Fill the two 2 KiB tables from the global W4096^m table: W256^(n k) as [k][n] (pass 2) and each thread's W4096^tid, the base
// of its pass-1 twiddles (they are its powers: pow_tree, as in fft_lds_core.h — one LDS read per frame and thread where two
// table reads and a product per twiddle were 30; parity unchanged, 162 -> 126 VGPRs in the flagship, DESIGN_APPENDIX.md A.13).
// Caller must __syncthreads() before the first f4k_transform().
__device__ __forceinline__ void f4k_init_tables(float2* __restrict__ tw256, float2* __restrict__ tw1,
                                                const float2* __restrict__ tw4096, int tid, F4kAddr& A) {
    (void)A;
    const int lo = tid & 15, hi = tid >> 4;
    tw256[tid] = tw4096[(16 * lo * hi) & (F4K_N - 1)];  // [k=hi][n=lo] = W256^(lo hi)
    tw1[tid] = tw4096[tid];                              // W4096^tid
}

// w[k] = w1^k for k = 1..15 with multiplication depth <= 4 (w2=w1^2, w4=w2^2, w8=w4^2).
__device__ __forceinline__ void pow_tree(cf w1, cf (&w)[16]) {
    w[1] = w1;
    w[2] = cmul(w1, w1);
    w[3] = cmul(w[2], w1);
    w[4] = cmul(w[2], w[2]);
    w[5] = cmul(w[4], w1);
    w[6] = cmul(w[4], w[2]);
    w[7] = cmul(w[4], w[3]);
    w[8] = cmul(w[4], w[4]);
    w[9] = cmul(w[8], w1);
    w[10] = cmul(w[8], w[2]);
    w[11] = cmul(w[8], w[3]);
    w[12] = cmul(w[8], w[4]);
    w[13] = cmul(w[8], w[5]);
    w[14] = cmul(w[8], w[6]);
    w[15] = cmul(w[8], w[7]);
}

// v[j] = x[tid + 256 j] on entry; on return X[tid + 256 k2] is in v[rev16(k2)].
// Contains four workgroup barriers; the first one also protects the previous
// call's exchange-2 reads, so calls may follow each other directly.
// With WIN, v[j] is still to be multiplied by the window coefficient win[j] = w[tid + 256 j] (folded into pass 1, cplx.h).
template <bool WIN = false>
__device__ __forceinline__ void f4k_transform(cf (&v)[16], float2* __restrict__ lds,
                                              const float2* __restrict__ tw256,
                                              const float2* __restrict__ tw1, const F4kAddr& A,
                                              int tid, const float* __restrict__ win = nullptr) {
    // ---- pass 1: DFT-16 over n2, times W4096^(r k0) = (W4096^r)^k0, r = tid ----
    radix16<WIN>(v, win);
    {
        const float2 t1 = tw1[tid];
        cf w[16];
        pow_tree(cf{t1.x, t1.y}, w);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[rev16(k)] = cmul(v[rev16(k)], w[k]);
    }
    __syncthreads();  // previous transform's pass-3 reads are done
#pragma unroll
    for (int k = 0; k < 16; ++k)
        lds[((k & 1) ? A.x1w_odd : A.x1w_even) + 256 * k] = make_float2(v[rev16(k)].x, v[rev16(k)].y);
    __syncthreads();
    // ---- pass 2: DFT-16 over n1 ; thread q = (n0=lo, k0=hi), times W256^(n0 k1) ----
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float2 t = lds[((j & 1) ? A.x1r_odd : A.x1r_even) + 16 * j];
        v[j] = cf{t.x, t.y};
    }
    radix16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        float2 w = tw256[16 * k + A.lo];
        v[rev16(k)] = cmul(v[rev16(k)], cf{w.x, w.y});
    }
    __syncthreads();  // exchange-1 reads are done
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[A.x2w + 16 * k] = make_float2(v[rev16(k)].x, v[rev16(k)].y);
    __syncthreads();
    // ---- pass 3: DFT-16 over n0 ; thread p = (k0=lo, k1=hi) = tid ----
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float2 t = lds[tid + 257 * j];
        v[j] = cf{t.x, t.y};
    }
    radix16(v);
}

}  // namespace sdrk
