// fft_lds32k.hip — frames of N = 32768 samples in ONE pass over HBM:
// window -> FFT -> fftshift -> 20*log10(|X|+eps)   (app/sdr/streamer.py:119,121 with rx_buffer_size = 2**15).
//
// A 32768-sample frame is 256 KiB of complex64 — more than the 160 KiB of LDS, which is why round 2 sent this
// length through the two tiled passes of fft_tiled2.hip (28 B/sample across the fabric, 0.31 of the HBM peak).  It
// still fits ONE CU if registers hold the frame and the LDS only carries the exchanges, one float plane at a time:
//   * 1024 threads, each plays two of the T = N/16 = 2048 "virtual threads" of the register + LDS construction of
//     fft_lds.hip (tau and tau + 1024): 32 points = 64 VGPRs of data per thread, N = 8 * 16 * 16 * 16;
//   * the three exchanges between the four passes go through a float[17/16 N] array (136 KiB) in two rounds —
//     real parts out and back, then imaginary parts — with the index rule of fft_lds_core.h unchanged (lane-
//     contiguous, bank-conflict free for 32-bit accesses as it was for 64-bit ones): 4 barriers per exchange.
// One workgroup per CU, no prefetch (the registers are full): load, transform and store phases of a CU do not
// overlap, so this runs at ~2/3 of what the smaller lengths reach — against the two-pass form's 0.31.
#include "fft_lds_core.h"

namespace sdrk {

namespace {
constexpr int L32 = 15;
using C32 = LdsCfg<L32>;
static_assert(C32::P == 4 && C32::R0 == 8 && C32::T == 2048, "decomposition of N = 32768");
constexpr int WG32 = 1024;

// one exchange: element (set s, slot k) of every thread goes to LDS index w[s][k]; afterwards register (s, j) holds the
// element at r[s][j].  Two rounds through the one float plane.
template <class WIdx, class RIdx>
__device__ __forceinline__ void exchange_split(cf (&v)[2][16], float* __restrict__ lds, WIdx widx, RIdx ridx) {
    float re[2][16];
    __syncthreads();                       // the previous exchange's last reads are done
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[widx(s, k)] = v[s][k].x;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 16; ++j) re[s][j] = lds[ridx(s, j)];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[widx(s, k)] = v[s][k].y;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[s][j] = cf{re[s][j], lds[ridx(s, j)]};
}
}  // namespace

template <bool HAS_WINDOW, int EPILOGUE>
__global__ __launch_bounds__(WG32, 4) void fft_lds32k_kernel(const float2* __restrict__ iq, size_t frame_stride,
                                                             void* __restrict__ out_raw, size_t n_frames,
                                                             const float* __restrict__ window,
                                                             const float2* __restrict__ twN /* W_N^m, m < N */, float eps,
                                                             int shift) {
    using C = C32;
    constexpr int N = C::N, T = C::T, R0 = C::R0, C0 = 16 / R0;
    extern __shared__ __attribute__((aligned(16))) float lds32k[];   // C::SLOT floats
    float* __restrict__ lds = lds32k;
    const int t = threadIdx.x;
    LdsTw<L32> tw[2];
    lds_tw_init<L32>(tw[0], twN, t);
    lds_tw_init<L32>(tw[1], twN, t + WG32);
    const int xor_q = shift ? 8 : 0;
    constexpr int OUT_ELEM = (EPILOGUE == EPI_LOGPSD ? 4 : 8);
    const __amdgpu_buffer_rsrc_t rwin = frame_rsrc(window, HAS_WINDOW ? N * 4 : 0);

    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rin = frame_rsrc(iq + f * frame_stride, N * 8);
        cf v[2][16];
        // register (s, i*R0 + j) <- sample n = tau_s + T (i + C0 j), tau_s = t + 1024 s
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const int step = WG32 * s + T * (i + C0 * j);
                    const v2f x = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rin, t * 8, step * 8, 2));
                    v[s][i * R0 + j] = cf{x.x, x.y};
                }
        if (HAS_WINDOW) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < C0; ++i)
#pragma unroll
                    for (int j = 0; j < R0; ++j) {
                        const int step = WG32 * s + T * (i + C0 * j);
                        const float w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rwin, t * 4, step * 4, 0));
                        v[s][i * R0 + j] = v[s][i * R0 + j] * w;
                    }
        }
        // ---- pass 0: two radix-8 butterflies per virtual thread, times W_N^((tau + T i) k) ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < C0; ++i) small_bfly<R0>(v[s], i * R0);
#pragma unroll
            for (int i = 0; i < C0; ++i) {
                cf w1 = tw[s].w0[i], wk = w1;
#pragma unroll
                for (int k = 1; k < R0; ++k) {
                    v[s][i * R0 + k] = cmul(v[s][i * R0 + k], wk);
                    if (k + 1 < R0) wk = cmul(wk, w1);
                }
            }
        }
        {   // exchange into the layout entering pass 1: slot i*R0 + k of virtual thread tau -> (tau + T i) + S1 k
            constexpr int S1 = C::Mp(0) + C::pad(1), Mq = C::Mp(1), Sin = C::Mp(0) + C::pad(1);
            exchange_split(v, lds,
                           [&](int s, int slot) { const int i = slot / R0, k = slot % R0; return (t + WG32 * s + T * i) + S1 * k; },
                           [&](int s, int j) { const int tau = t + WG32 * s, Kin = tau / Mq, rr = tau - Kin * Mq; return rr + Mq * j + Sin * Kin; });
        }
        // ---- passes 1 and 2: radix 16, twiddle, exchange ----
#pragma unroll
        for (int p = 1; p <= 2; ++p) {
            const int Mq = C::Mp(p);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                radix16(v[s]);
                cf w[16], w1 = tw[s].wp[p];
                asm volatile("" : "+v"(w1.x), "+v"(w1.y));   // keep the power tree inside the frame loop
                pow_tree(w1, w);
                cf o[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) o[k] = k ? cmul(v[s][rev16(k)], w[k]) : v[s][rev16(0)];   // natural order k
#pragma unroll
                for (int k = 0; k < 16; ++k) v[s][k] = o[k];
            }
            const int Sout = Mq + C::pad(p + 1), kstep = N / C::Np(p);
            const int Mn = C::Mp(p + 1), Sin = C::Mp(p) + C::pad(p + 1);
            exchange_split(v, lds,
                           [&](int s, int k) { const int tau = t + WG32 * s, Kin = tau / Mq, rr = tau - Kin * Mq; return rr + Sout * (Kin + kstep * k); },
                           [&](int s, int j) { const int tau = t + WG32 * s, Kin = tau / Mn, rr = tau - Kin * Mn; return rr + Mn * j + Sin * Kin; });
        }
        // ---- pass 3 and the epilogue: X[tau + T q] is in v[s][rev16(q)] ----
        const __amdgpu_buffer_rsrc_t w = frame_rsrc(static_cast<char*>(out_raw) + f * (size_t)N * OUT_ELEM, (unsigned)(N * OUT_ELEM));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            radix16(v[s]);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[s][rev16(q)];
                const int step = (WG32 * s + T * (q ^ xor_q)) * OUT_ELEM;
                if (EPILOGUE == EPI_LOGPSD) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, logpsd_db(z.x, z.y, eps)), w, t * OUT_ELEM, step, 2);
                } else {
                    const v2f o = {z.x, z.y};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), w, t * OUT_ELEM, step, 0);
                }
            }
        }
    }
}

hipError_t launch_fft_lds32k(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    if (a.nfft != C32::N) return hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)C32::SLOT * sizeof(float);
    const unsigned grid = (unsigned)(a.n_frames < (size_t)a.num_cus ? a.n_frames : (size_t)a.num_cus);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
#define SDRK_32K(W, E)                                                                                        \
    do {                                                                                                      \
        auto kern = fft_lds32k_kernel<W, E>;                                                                  \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                  \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);           \
        if (e0 != hipSuccess) return e0;                                                                      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WG32), lds_bytes, a.stream, iq, a.frame_stride, a.d_out,    \
                           a.n_frames, a.d_window, tw, a.eps, a.shift);                                       \
    } while (0)
    if (a.epilogue == EPI_LOGPSD) {
        if (a.d_window) SDRK_32K(true, EPI_LOGPSD); else SDRK_32K(false, EPI_LOGPSD);
    } else {
        if (a.d_window) SDRK_32K(true, EPI_COMPLEX); else SDRK_32K(false, EPI_COMPLEX);
    }
#undef SDRK_32K
    return hipGetLastError();
}

}  // namespace sdrk
