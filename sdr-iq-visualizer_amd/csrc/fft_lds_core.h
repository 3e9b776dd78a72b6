// fft_lds_core.h — the register + LDS transform shared by fft_lds.hip (whole frames of 16…16384
// samples) and fft_tiled2.hip (column / row passes of the two-pass large-frame path).
// See fft_lds.hip for the decomposition, the LDS layout rule and the twiddle scheme.
#pragma once
#include "fft4096_core.h"

namespace sdrk {

template <int LOG2N>
struct LdsCfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int P = (LOG2N + 3) / 4;               // passes
    static constexpr int R0 = 1 << (LOG2N - 4 * (P - 1));   // first radix
    static constexpr int T = N / 16;                        // threads per frame
    static constexpr int WG = T < 256 ? 256 : T;
    static constexpr int F = WG / T;                        // frames per workgroup pass
    static constexpr int SLOT = N + N / 16;                 // LDS elements per frame (17/16 N)
    static constexpr bool PREFETCH = WG <= 256;   // wider workgroups are capped at 128 VGPRs
    static constexpr int WAVES = WG == 1024 ? 4 : (WG == 512 ? 4 : 3);  // waves/SIMD asked of the compiler
    __host__ __device__ static constexpr int radix(int p) { return p == 0 ? R0 : 16; }
    __host__ __device__ static constexpr int Np(int p) { return p == 0 ? N : (N / R0) >> (4 * (p - 1)); }
    __host__ __device__ static constexpr int Mp(int p) { return Np(p) / radix(p); }
    __host__ __device__ static constexpr int pad(int p) { return (p < P && Mp(p) < 32) ? Mp(p) : 0; }  // pad of the layout entering pass p
};

// small first-pass butterflies on v[base .. base+R)
template <int R>
__device__ __forceinline__ void small_bfly(cf (&v)[16], int base) {
    if (R == 2) bfly2(v[base], v[base + 1]);
    if (R == 4) bfly4(v[base], v[base + 1], v[base + 2], v[base + 3]);
    if (R == 8) {
        cf e0 = v[base], e1 = v[base + 2], e2 = v[base + 4], e3 = v[base + 6];
        cf o0 = v[base + 1], o1 = v[base + 3], o2 = v[base + 5], o3 = v[base + 7];
        bfly4(e0, e1, e2, e3);
        bfly4(o0, o1, o2, o3);
        cf t1 = rot_m45(o1);
        cf t2 = mul_mi(o2);
        cf t3 = rot_m135(o3);
        v[base] = e0 + o0; v[base + 4] = e0 - o0;
        v[base + 1] = e1 + t1; v[base + 5] = e1 - t1;
        v[base + 2] = e2 + t2; v[base + 6] = e2 - t2;
        v[base + 3] = e3 + t3; v[base + 7] = e3 - t3;
    }
}

// Per-thread twiddle bases (constant across frames): pass 0 needs W_N^(tau + T i), pass p >= 1 needs
// W_{N_p}^(tau % M_p); both come from the W_N table of the transform length.
template <int LOG2N>
struct LdsTw {
    cf w0[16 / LdsCfg<LOG2N>::R0];
    cf wp[LdsCfg<LOG2N>::P > 1 ? LdsCfg<LOG2N>::P : 1];
};

template <int LOG2N>
__device__ __forceinline__ void lds_tw_init(LdsTw<LOG2N>& tw, const float2* __restrict__ twN, int tau) {
    using C = LdsCfg<LOG2N>;
#pragma unroll
    for (int i = 0; i < 16 / C::R0; ++i) {
        const float2 t = twN[(tau + C::T * i) & (C::N - 1)];
        tw.w0[i] = cf{t.x, t.y};
    }
#pragma unroll
    for (int p = 1; p < C::P - 1; ++p) {
        const float2 t = twN[((tau % C::Mp(p)) * (C::N / C::Np(p))) & (C::N - 1)];
        tw.wp[p] = cf{t.x, t.y};
    }
}

// compiler fence beside the wave barriers of the barrier-free exchanges (SDRK_WAVE_SYNC_CLOBBER=0 in A/B builds: round 4's form)
#ifndef SDRK_WAVE_SYNC_CLOBBER
#define SDRK_WAVE_SYNC_CLOBBER 1
#endif
#if SDRK_WAVE_SYNC_CLOBBER
#define SDRK_WAVE_SYNC_FENCE() asm volatile("" ::: "memory")
#else
#define SDRK_WAVE_SYNC_FENCE() do { } while (0)
#endif

// Workgroup barrier of the transform.  RAW = false: __syncthreads().  RAW = true: wait for this wave's own LDS
// operations only, then s_barrier — for callers that keep an LDS-DMA (`buffer_load ... lds`) in flight across the
// transform into a DIFFERENT part of LDS: __syncthreads() would make hipcc wait vmcnt(0) at every barrier and
// drain it.  The empty asm statements keep the compiler from moving LDS accesses across the barrier.
// SYNC = 2: no barrier at all — for callers whose frame (its T threads) lies inside ONE wave (T <= 64, IL = 1 with a
// per-frame base): the wave's own LDS queue orders its exchanges; only the compiler is kept from moving LDS accesses.
template <int SYNC>
__device__ __forceinline__ void lds_core_barrier() {
    if (SYNC == 2) {
        // (the intrinsic is a scheduling barrier with side effects, not a memory clobber: the empty asm statements are what
        //  forbids the compiler to move one lane's LDS store below — or its next LDS load above — the other lanes' exchange)
        SDRK_WAVE_SYNC_FENCE();
        __builtin_amdgcn_wave_barrier();
        SDRK_WAVE_SYNC_FENCE();
    } else if (SYNC == 1) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); vmcnt and expcnt untouched
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
}

// The transform of one frame by T = N/16 threads, 16 points each.  v[i*R0 + j] = x[tau + T (i + C0 j)] on entry
// (C0 = 16/R0); on return X[tau + T q] is in v[rev16(q)] (for P == 1, i.e. N == 16: X[k] in v[rev16(k)]).  LDS
// element (inner address a) lives at lds[a * IL + off]: IL = 1 with a per-frame base for frame-per-thread-group
// use, IL = W and off = column for W interleaved columns.  Contains 2 (P-1) workgroup barriers; the first also
// protects the previous call's last reads.  SYNC: 0 = __syncthreads, 1 = raw s_barrier, 2 = none (see lds_core_barrier).
template <int LOG2N, int IL, int SYNC = 0>
__device__ __forceinline__ void lds_fft_core(cf (&v)[16], float2* __restrict__ lds, int off, int tau,
                                             const LdsTw<LOG2N>& tw) {
    using C = LdsCfg<LOG2N>;
    constexpr int N = C::N, P = C::P, R0 = C::R0, T = C::T;
    constexpr int C0 = 16 / R0;
    if (R0 == 16) {
        radix16(v);
    } else {
#pragma unroll
        for (int i = 0; i < C0; ++i) small_bfly<R0>(v, i * R0);
    }
    if (P > 1) {
        if (R0 == 16) {
            cf w[16], w1 = tw.w0[0];
            asm volatile("" : "+v"(w1.x), "+v"(w1.y));  // keep the tree inside the frame loop (no LICM into 30 VGPRs)
            pow_tree(w1, w);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[rev16(k)] = cmul(v[rev16(k)], w[k]);
        } else {
#pragma unroll
            for (int i = 0; i < C0; ++i) {
                cf w1 = tw.w0[i], wk = w1;
#pragma unroll
                for (int k = 1; k < R0; ++k) {
                    v[i * R0 + k] = cmul(v[i * R0 + k], wk);
                    if (k + 1 < R0) wk = cmul(wk, w1);
                }
            }
        }
        lds_core_barrier<SYNC>();  // previous transform's last-pass reads are done
        constexpr int S1 = C::Mp(0) + C::pad(1);
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int k = 0; k < R0; ++k) {
                const cf z = v[i * R0 + (R0 == 16 ? rev16(k) : k)];
                lds[((tau + T * i) + S1 * k) * IL + off] = make_float2(z.x, z.y);
            }
        lds_core_barrier<SYNC>();
    }
#pragma unroll
    for (int p = 1; p < P; ++p) {
        const int Mq = C::Mp(p);
        const int Sin = C::Mp(p - 1) + C::pad(p);
        const int Kin = tau / Mq, rr = tau - Kin * Mq;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float2 t = lds[(rr + Mq * j + Sin * Kin) * IL + off];
            v[j] = cf{t.x, t.y};
        }
        radix16(v);
        if (p < P - 1) {
            cf w[16], w1 = tw.wp[p];
            asm volatile("" : "+v"(w1.x), "+v"(w1.y));
            pow_tree(w1, w);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[rev16(k)] = cmul(v[rev16(k)], w[k]);
            lds_core_barrier<SYNC>();  // everyone has read the layout entering pass p
            const int Sout = Mq + C::pad(p + 1);
            const int kstep = N / C::Np(p);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const cf z = v[rev16(k)];
                lds[(rr + Sout * (Kin + kstep * k)) * IL + off] = make_float2(z.x, z.y);
            }
            lds_core_barrier<SYNC>();
        }
    }
}


// TWO threads of the transform per lane (taua and taub = taua + T/2: 32 points in 64 registers), for frames whose T/2
// lanes are ONE wave (T = 128: rows of 2048 points): the exchanges then need no barrier at all — the wave's own LDS queue is
// the ordering, __builtin_amdgcn_wave_barrier() only keeps the compiler from moving LDS accesses across — and a kernel of
// such waves runs at two per SIMD, 256 registers each (room for parked results).  The second thread's first-pass table
// entries are W_N^(T/2) = W_32 times the first's and its later passes' entries coincide with the first's (M_p divides T/2):
// nv2_tw_b.  (The same two-threads-per-lane form WITH workgroup barriers, for frames that span several waves — N = 8192 /
// 16384 at 256 / 512 lanes with the next frame prefetched into the spare registers — was built and measured in round 4:
// parity-green and 15 / 25 % SLOWER than the 512 / 1024-thread kernels, 0.340 against 0.296 and 0.44 against 0.352 ms per
// 2^27 samples.  Half the waves hide half the latency; the barrier-free case is the one that pays.)
__device__ __forceinline__ void nv2_sync() {
    SDRK_WAVE_SYNC_FENCE();                  // (see lds_core_barrier<2>)
    __builtin_amdgcn_wave_barrier();
    SDRK_WAVE_SYNC_FENCE();
}

template <int LOG2N>
__device__ __forceinline__ LdsTw<LOG2N> nv2_tw_b(const LdsTw<LOG2N>& twa) {
    LdsTw<LOG2N> twb = twa;
    const cf w32 = cf{0.98078528040323044913f, -0.19509032201612826785f};   // W_N^(N/32) = exp(-i pi / 16)
#pragma unroll
    for (int i = 0; i < 16 / LdsCfg<LOG2N>::R0; ++i) twb.w0[i] = cmul(twa.w0[i], w32);
    return twb;
}

template <int LOG2N>
__device__ __forceinline__ void lds_fft_core_nv2(cf (&va)[16], cf (&vb)[16], float2* __restrict__ lds, int taua, int taub,
                                                   const LdsTw<LOG2N>& twa, const LdsTw<LOG2N>& twb) {
    using C = LdsCfg<LOG2N>;
    constexpr int N = C::N, P = C::P, R0 = C::R0, T = C::T, C0 = 16 / R0;
    static_assert(R0 != 16 && P > 1, "written for the R0 < 16 first pass (2048 = 8 x 16 x 16, 8192 = 2 x 16^3, 16384 = 4 x 16^3)");
    auto first = [&](cf (&v)[16], const LdsTw<LOG2N>& tw) {
#pragma unroll
        for (int i = 0; i < C0; ++i) small_bfly<R0>(v, i * R0);
#pragma unroll
        for (int i = 0; i < C0; ++i) {
            cf w1 = tw.w0[i], wk = w1;
#pragma unroll
            for (int k = 1; k < R0; ++k) {
                v[i * R0 + k] = cmul(v[i * R0 + k], wk);
                if (k + 1 < R0) wk = cmul(wk, w1);
            }
        }
    };
    first(va, twa);
    first(vb, twb);
    nv2_sync();
    constexpr int S1 = C::Mp(0) + C::pad(1);
    auto write1 = [&](const cf (&v)[16], int tau) {
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int k = 0; k < R0; ++k) {
                const cf z = v[i * R0 + k];
                lds[(tau + T * i) + S1 * k] = make_float2(z.x, z.y);
            }
    };
    write1(va, taua);
    write1(vb, taub);
#pragma unroll
    for (int p = 1; p < P; ++p) {
        const int Mq = C::Mp(p);
        const int Sin = C::Mp(p - 1) + C::pad(p);
        auto read = [&](cf (&v)[16], int tau) {
            const int Kin = tau / Mq, rr = tau - Kin * Mq;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 t = lds[rr + Mq * j + Sin * Kin];
                v[j] = cf{t.x, t.y};
            }
        };
        nv2_sync();                                // the layout entering pass p is complete
        read(va, taua);
        read(vb, taub);
        radix16(va);
        radix16(vb);
        if (p < P - 1) {
            auto twiddle = [&](cf (&v)[16], const LdsTw<LOG2N>& tw) {
                cf w[16], w1 = tw.wp[p];
                asm volatile("" : "+v"(w1.x), "+v"(w1.y));
                pow_tree(w1, w);
#pragma unroll
                for (int k = 1; k < 16; ++k) v[rev16(k)] = cmul(v[rev16(k)], w[k]);
            };
            twiddle(va, twa);
            __builtin_amdgcn_sched_barrier(0);     // (one power tree at a time: interleaved, the two cost 32 more registers)
            twiddle(vb, twb);
            const int Sout = Mq + C::pad(p + 1);
            const int kstep = N / C::Np(p);
            auto write = [&](const cf (&v)[16], int tau) {
                const int Kin = tau / Mq, rr = tau - Kin * Mq;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const cf z = v[rev16(k)];
                    lds[rr + Sout * (Kin + kstep * k)] = make_float2(z.x, z.y);
                }
            };
            nv2_sync();                            // both of the lane's threads have read the layout entering pass p
            write(va, taua);
            write(vb, taub);
        }
    }
}


// Scratch layout between the passes.  The col pass produces, per workgroup, all A values of k3 for W adjacent
// m; the row pass consumes 16 adjacent k3 for all M values of m: whatever the layout, the two footprints meet
// in 16 x 16 element squares.  Stored as [k3/16][m/16][k3%16][m%16] (2 KiB squares, row tiles contiguous)
// the col pass writes 2 KiB runs (512 B per wave instruction) instead of the 128-byte pieces at M*8-byte
// stride a plain [k3][m] matrix costs it, and the row pass reads its 16 x M tile as one contiguous block.
// element index of (k3, m) inside one frame's scratch
__device__ __forceinline__ int scratch_index(int k3, int m, int M) {
    return (((k3 >> 4) * (M >> 4) + (m >> 4)) << 8) + ((k3 & 15) << 4) + (m & 15);
}

}  // namespace sdrk
