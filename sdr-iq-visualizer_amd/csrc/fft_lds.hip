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
#include "fft_lds_core.h"

namespace sdrk {

// STAGED (N <= 64, packed frames only): with fewer than 8 threads per frame the per-thread pattern
// x[tau + T q] touches 64 different 128-byte lines per wave instruction, so the group's 4096 contiguous
// samples are loaded lane-contiguously, parked in LDS (padded by one element per 16) and picked up from
// there; the rows take the same route out.
template <int LOG2N, bool HAS_WINDOW, int EPILOGUE, bool STAGED>
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

    LdsTw<LOG2N> tw;
    lds_tw_init<LOG2N>(tw, twN, tau);

    const size_t n_groups = (n_frames + F - 1) / F;
    const int xor_q = shift ? 8 : 0;

    // Buffer addressing: a wave-uniform descriptor on the group's F frames, clipped to the frames that
    // exist (lanes of missing frames read zeros and their stores are dropped by the bounds check), one
    // 32-bit lane offset, uniform steps of T elements.
    const int lane_in = STAGED ? tid * 8 : (int)((size_t)fr * frame_stride + tau) * 8;
    constexpr int OUT_ELEM = (EPILOGUE == EPI_LOGPSD ? 4 : 8);
    const int lane_out = STAGED ? tid * OUT_ELEM : (fr * N + tau) * OUT_ELEM;
    constexpr int IN_STEP = STAGED ? 256 : T;   // elements between a thread's consecutive loads
    v2u nxt[16];
    auto issue_loads = [&](size_t g) {
        const size_t f0 = g * F;
        const size_t valid = n_frames - f0 < (size_t)F ? n_frames - f0 : (size_t)F;
        const __amdgpu_buffer_rsrc_t r = frame_rsrc(iq + f0 * frame_stride, (unsigned)(((valid - 1) * frame_stride + N) * 8));
#pragma unroll
        for (int q = 0; q < 16; ++q) nxt[q] = __builtin_amdgcn_raw_buffer_load_b64(r, lane_in, q * IN_STEP * 8, 2);
    };
    if (C::PREFETCH && blockIdx.x < n_groups) issue_loads(blockIdx.x);

    for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        if (!C::PREFETCH) issue_loads(g);
        cf v[16];
        if constexpr (STAGED) {
            // park element e = tid + 256 q of the group (frame e / N, sample e % N) ...
            __syncthreads();   // the previous group's staged rows have been read back
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + 256 * q, fe = e / N, n = e - fe * N;
                const v2f t = __builtin_bit_cast(v2f, nxt[q]);
                lds_all[fe * C::SLOT + n + (n >> 4)] = make_float2(t.x, t.y);
            }
            __syncthreads();
            // ... and pick up register i*R0 + j <- sample n = tau + T (i + C0 j) of this thread's frame
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const int n = tau + T * (i + C0 * j);
                    const float2 t = lds[n + (n >> 4)];
                    v[i * R0 + j] = cf{t.x, t.y};
                }
            if (P == 1) __syncthreads();   // (for P > 1 the transform's first barrier covers these reads)
        } else {
            // register i*R0 + j <- sample n = tau + T i + M_0 j = tau + T (i + C0 j)
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const v2f t = __builtin_bit_cast(v2f, nxt[i + C0 * j]);
                    v[i * R0 + j] = cf{t.x, t.y};
                }
        }
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

        lds_fft_core<LOG2N, 1>(v, lds, 0, tau, tw);
        // ---------------- epilogue: X[tau + T q] (for N == 16: X[k]) ----------------
        if constexpr (STAGED) {
            const size_t f0 = g * F;
            const size_t valid = n_frames - f0 < (size_t)F ? n_frames - f0 : (size_t)F;
            const __amdgpu_buffer_rsrc_t w = frame_rsrc(static_cast<char*>(out_raw) + f0 * (size_t)N * OUT_ELEM,
                                                        (unsigned)(valid * N * OUT_ELEM));
            __syncthreads();   // every thread is through its last exchange reads: LDS becomes the row staging area
            if (EPILOGUE == EPI_LOGPSD) {
                float* __restrict__ st = reinterpret_cast<float*>(lds_all);   // frame fe at fe * (N + 1)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const cf z = v[rev16(q)];
                    const int k = P == 1 ? (q ^ xor_q) : tau + T * (q ^ xor_q);
                    st[fr * (N + 1) + k] = logpsd_db(z.x, z.y, eps);
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int e = tid + 256 * q, fe = e / N, k = e - fe * N;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, st[fe * (N + 1) + k]), w, lane_out,
                                                          q * 256 * OUT_ELEM, 2);
                }
            } else {
                float2* __restrict__ st = lds_all;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const cf z = v[rev16(q)];
                    const int k = P == 1 ? (q ^ xor_q) : tau + T * (q ^ xor_q);
                    st[fr * (N + 1) + k] = make_float2(z.x, z.y);
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int e = tid + 256 * q, fe = e / N, k = e - fe * N;
                    const float2 t = st[fe * (N + 1) + k];
                    const v2f o = {t.x, t.y};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), w, lane_out, q * 256 * OUT_ELEM, 0);
                }
            }
        } else {
            const size_t f0 = g * F;
            const size_t valid = n_frames - f0 < (size_t)F ? n_frames - f0 : (size_t)F;
            const __amdgpu_buffer_rsrc_t w = frame_rsrc(static_cast<char*>(out_raw) + f0 * (size_t)N * OUT_ELEM,
                                                        (unsigned)(valid * N * OUT_ELEM));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[rev16(q)];
                const int step = (P == 1 ? (q ^ xor_q) : T * (q ^ xor_q)) * OUT_ELEM;
                if (EPILOGUE == EPI_LOGPSD) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, logpsd_db(z.x, z.y, eps)), w,
                                                          lane_out, step, 2);
                } else {
                    const v2f o = {z.x, z.y};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), w, lane_out, step, 0);
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
    const bool staged = LOG2N <= 7 && a.frame_stride == (size_t)C::N;
#define SDRK_LDS(W, E)                                                                                        \
    do {                                                                                                      \
        auto kern = (LOG2N <= 7 && staged) ? fft_lds_kernel<LOG2N, W, E, (LOG2N <= 7)>                        \
                                            : fft_lds_kernel<LOG2N, W, E, false>;                             \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                  \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);           \
        if (e0 != hipSuccess) return e0;                                                                      \
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

bool fft_lds_supports(int nfft) { return nfft >= 16 && nfft <= 16384 && nfft != 4096 && (nfft & (nfft - 1)) == 0; }

hipError_t launch_fft_lds(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    // Frames of one workgroup share a buffer descriptor with 32-bit lane offsets: absurdly spaced frames
    // (group span >= 2 GiB) go to the pointer-addressed catch-all instead.
    if (a.nfft < 4096 && ((size_t)(4096 / a.nfft) * a.frame_stride + (size_t)a.nfft) * 8 >= ((size_t)1 << 31))
        return launch_fft_small(a);
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
