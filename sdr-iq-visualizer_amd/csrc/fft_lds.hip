// fft_lds.hip — frames of N = 16 … 16384 samples (other than the 4096 of fft4096.hip), the
// whole transform in registers + LDS, one HBM read and one HBM write per sample:
// window -> FFT -> fftshift -> 20*log10(|X|+eps)   (app/sdr/streamer.py:119,121 with a
// non-default rx_buffer_size, streamer.py:10).
//
// Same construction as the flagship kernel, generalised: 16 points per thread, N/16 threads
// per frame, passes of radix R0 (2, 4, 8 or 16; first) then 16, 16, …; between passes one LDS
// exchange.  N = R0 * 16^(P-1):
//   state before pass p: (K, r), K = k0 + R0 k1 + … (outputs so far), r in [0, N_p), N_p = N / (R0…R_{p-1})
//   pass p butterfly (K, r'), r' < M_p = N_p / R_p : inputs r = r' + M_p j ; output k_p -> (K + (N/N_p) k_p, r'),
//   times W_{N_p}^(r' k_p) unless it is the last pass; after the last pass X[K].
// LDS address of state (K, r) entering pass p+1:  r + (M_p + pad) K  with pad = M_{p+1} when
// M_{p+1} < 32 (else 0): writes (lanes along r') and reads (lanes along r'' then K) are both
// bank-conflict free for 64-bit accesses; a frame needs 17/16 N complex64 of LDS.
// Loads x[tau + (N/16) q] and stores out[tau + (N/16) q] are lane-contiguous.  Small frames share
// a 256-thread workgroup (4096/N frames at a time); N = 8192 / 16384 use 512 / 1024 threads.
// Twiddles: one table entry per thread and pass (constant across frames, kept in registers) and a
// depth<=4 product tree for its powers.  Persistent grid; the next group's loads are prefetched
// into registers while the current one is transformed (except at 1024 threads, for VGPRs).
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
        constexpr float R2 = 0.70710678118654752440f;
        cf e0 = v[base], e1 = v[base + 2], e2 = v[base + 4], e3 = v[base + 6];
        cf o0 = v[base + 1], o1 = v[base + 3], o2 = v[base + 5], o3 = v[base + 7];
        bfly4(e0, e1, e2, e3);
        bfly4(o0, o1, o2, o3);
        cf t1 = cf{(o1.x + o1.y) * R2, (o1.y - o1.x) * R2};
        cf t2 = mul_mi(o2);
        cf t3 = cf{(o3.y - o3.x) * R2, -(o3.x + o3.y) * R2};
        v[base] = e0 + o0; v[base + 4] = e0 - o0;
        v[base + 1] = e1 + t1; v[base + 5] = e1 - t1;
        v[base + 2] = e2 + t2; v[base + 6] = e2 - t2;
        v[base + 3] = e3 + t3; v[base + 7] = e3 - t3;
    }
}

template <int LOG2N, bool HAS_WINDOW, int EPILOGUE>
__global__ __launch_bounds__(LdsCfg<LOG2N>::WG, LdsCfg<LOG2N>::WAVES) void fft_lds_kernel(
    const float2* __restrict__ iq, size_t frame_stride, void* __restrict__ out_raw, size_t n_frames,
    const float* __restrict__ window, const float2* __restrict__ twN /* W_N^m, m < N */, float eps, int shift) {
    using C = LdsCfg<LOG2N>;
    constexpr int N = C::N, P = C::P, R0 = C::R0, T = C::T, F = C::F;
    constexpr int C0 = 16 / R0;  // butterflies per thread in pass 0
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];

    const int tid = threadIdx.x;
    const int fr = tid / T, tau = tid - fr * T;
    float2* __restrict__ lds = lds_all + (size_t)fr * C::SLOT;

    // per-thread twiddle bases, constant across frames
    cf w0[C0];        // pass 0: W_N^(r'), r' = tau + T i
#pragma unroll
    for (int i = 0; i < C0; ++i) {
        const float2 t = twN[(tau + T * i) & (N - 1)];
        w0[i] = cf{t.x, t.y};
    }
    cf wp[P > 1 ? P : 1];  // passes 1..P-2: W_{N_p}^(r''), r'' = tau % M_p
#pragma unroll
    for (int p = 1; p < P - 1; ++p) {
        const float2 t = twN[((tau % C::Mp(p)) * (N / C::Np(p))) & (N - 1)];
        wp[p] = cf{t.x, t.y};
    }

    const size_t n_groups = (n_frames + F - 1) / F;
    const int xor_q = shift ? 8 : 0;

    v2f nxt[16];
    auto issue_loads = [&](size_t g) {
        const size_t f = g * F + fr;
        const bool ok = f < n_frames;
        const v2f* __restrict__ x = reinterpret_cast<const v2f*>(iq) + (ok ? f : 0) * frame_stride + tau;
#pragma unroll
        for (int q = 0; q < 16; ++q) nxt[q] = ok ? __builtin_nontemporal_load(&x[T * q]) : v2f{0.f, 0.f};
    };
    if (C::PREFETCH && blockIdx.x < n_groups) issue_loads(blockIdx.x);

    for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        if (!C::PREFETCH) issue_loads(g);
        cf v[16];
        // register i*R0 + j <- sample n = tau + T i + M_0 j = tau + T (i + C0 j)
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) v[i * R0 + j] = cf{nxt[i + C0 * j].x, nxt[i + C0 * j].y};
        if (C::PREFETCH) {
            const size_t gn = g + gridDim.x;
            issue_loads(gn < n_groups ? gn : g);
        }
        if (HAS_WINDOW) {
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) v[i * R0 + j] = v[i * R0 + j] * window[tau + T * (i + C0 * j)];
        }

        // ---------------- pass 0 ----------------
        if (R0 == 16) {
            radix16(v);
        } else {
#pragma unroll
            for (int i = 0; i < C0; ++i) small_bfly<R0>(v, i * R0);
        }
        if (P > 1) {
            // twiddle W_N^(r' k0) and scatter to the layout entering pass 1: r' + (M_0 + pad_1) * k0
            if (R0 == 16) {
                cf w[16], w1 = w0[0];
                asm volatile("" : "+v"(w1.x), "+v"(w1.y));
                pow_tree(w1, w);
#pragma unroll
                for (int k = 1; k < 16; ++k) v[rev16(k)] = cmul(v[rev16(k)], w[k]);
            } else {
#pragma unroll
                for (int i = 0; i < C0; ++i) {
                    cf w1 = w0[i], wk = w1;
#pragma unroll
                    for (int k = 1; k < R0; ++k) {
                        v[i * R0 + k] = cmul(v[i * R0 + k], wk);
                        if (k + 1 < R0) wk = cmul(wk, w1);
                    }
                }
            }
            __syncthreads();  // previous group's last-pass reads are done
            constexpr int S1 = C::Mp(0) + C::pad(1);
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int k = 0; k < R0; ++k) {
                    const cf z = v[i * R0 + (R0 == 16 ? rev16(k) : k)];
                    lds[(tau + T * i) + S1 * k] = make_float2(z.x, z.y);
                }
            __syncthreads();
        }
        // ---------------- passes 1 .. P-1 (radix 16, one butterfly per thread) ----------------
#pragma unroll
        for (int p = 1; p < P; ++p) {
            const int Mq = C::Mp(p);                       // M_p
            const int Sin = C::Mp(p - 1) + C::pad(p);      // K stride of the layout entering pass p
            const int Kin = tau / Mq, rr = tau - Kin * Mq; // butterfly (K, r'')
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 t = lds[rr + Mq * j + Sin * Kin];
                v[j] = cf{t.x, t.y};
            }
            radix16(v);
            if (p < P - 1) {
                cf w[16], w1 = wp[p];
                asm volatile("" : "+v"(w1.x), "+v"(w1.y));
                pow_tree(w1, w);
#pragma unroll
                for (int k = 1; k < 16; ++k) v[rev16(k)] = cmul(v[rev16(k)], w[k]);
                __syncthreads();  // everyone has read the layout entering pass p
                const int Sout = Mq + C::pad(p + 1);
                const int kstep = N / C::Np(p);            // K_{p+1} = K + kstep * k_p
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const cf z = v[rev16(k)];
                    lds[rr + Sout * (Kin + kstep * k)] = make_float2(z.x, z.y);
                }
                __syncthreads();
            }
        }
        // ---------------- epilogue: X[K + (N/16) q] for K = tau (P > 1) ----------------
        const size_t f = g * F + fr;
        if (f < n_frames) {
            if (P == 1) {
                // N == 16: the single butterfly's outputs are the spectrum
                if (EPILOGUE == EPI_LOGPSD) {
                    float* __restrict__ o = static_cast<float*>(out_raw) + f * (size_t)N;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { cf z = v[rev16(k)]; o[k ^ xor_q] = logpsd_db(z.x, z.y, eps); }
                } else {
                    float2* __restrict__ o = static_cast<float2*>(out_raw) + f * (size_t)N;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { cf z = v[rev16(k)]; o[k ^ xor_q] = make_float2(z.x, z.y); }
                }
            } else if (EPILOGUE == EPI_LOGPSD) {
                float* __restrict__ o = static_cast<float*>(out_raw) + f * (size_t)N + tau;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    cf z = v[rev16(q)];
                    __builtin_nontemporal_store(logpsd_db(z.x, z.y, eps), &o[T * (q ^ xor_q)]);
                }
            } else {
                float2* __restrict__ o = static_cast<float2*>(out_raw) + f * (size_t)N + tau;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    cf z = v[rev16(q)];
                    o[T * (q ^ xor_q)] = make_float2(z.x, z.y);
                }
            }
        }
    }
}

template <int LOG2N>
static hipError_t launch_lds_n(const LaunchArgs& a) {
    using C = LdsCfg<LOG2N>;
    const size_t n_groups = (a.n_frames + C::F - 1) / C::F;
    const size_t lds_bytes = (size_t)C::F * C::SLOT * sizeof(float2);
    size_t per_cu = (160 * 1024) / (lds_bytes ? lds_bytes : 1);
    const size_t by_threads = 2048 / C::WG;   // 32 waves per CU
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const size_t max_blocks = (size_t)a.num_cus * per_cu;
    const unsigned grid = (unsigned)(n_groups < max_blocks ? n_groups : max_blocks);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
#define SDRK_LDS(W, E)                                                                                        \
    do {                                                                                                      \
        auto kern = fft_lds_kernel<LOG2N, W, E>;                                                              \
        if (lds_bytes > 64 * 1024) {                                                                          \
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
            if (e0 != hipSuccess) return e0;                                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::WG), lds_bytes, a.stream, iq, a.frame_stride, a.d_out,   \
                           a.n_frames, a.d_window, tw, a.eps, a.shift);                                       \
    } while (0)
    if (a.epilogue == EPI_LOGPSD) {
        if (a.d_window) SDRK_LDS(true, EPI_LOGPSD); else SDRK_LDS(false, EPI_LOGPSD);
    } else {
        if (a.d_window) SDRK_LDS(true, EPI_COMPLEX); else SDRK_LDS(false, EPI_COMPLEX);
    }
#undef SDRK_LDS
    return hipGetLastError();
}

bool fft_lds_supports(int nfft) { return nfft >= 16 && nfft <= 16384 && (nfft & (nfft - 1)) == 0 && nfft != 4096; }

hipError_t launch_fft_lds(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    switch (a.nfft) {
        case 16: return launch_lds_n<4>(a);
        case 32: return launch_lds_n<5>(a);
        case 64: return launch_lds_n<6>(a);
        case 128: return launch_lds_n<7>(a);
        case 256: return launch_lds_n<8>(a);
        case 512: return launch_lds_n<9>(a);
        case 1024: return launch_lds_n<10>(a);
        case 2048: return launch_lds_n<11>(a);
        case 8192: return launch_lds_n<13>(a);
        case 16384: return launch_lds_n<14>(a);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sdrk
