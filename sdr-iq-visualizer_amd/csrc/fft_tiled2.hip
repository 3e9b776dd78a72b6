// fft_tiled2.hip — frames of N = A * M samples, 2^15 <= N <= 2^20, in TWO tiled passes whose
// sub-transforms (A, M <= 1024 points) run entirely in registers + LDS (fft_lds_core.h):
// window -> FFT -> fftshift -> 20*log10(|X|+eps)   (app/sdr/streamer.py:119,121); the
// waterfall sizes of BASELINE.json configs 3 (N = 65536) and 5 (N = 2^20).
//
//   n = m + M n3 (m < M contiguous, n3 < A),   k = A km + k3
//   X[A km + k3] = sum_m W_M^(m km) * W_N^(m k3) * [ sum_n3 x[m + M n3] W_A^(n3 k3) ]
//
//   col pass   tile = 16 adjacent m x all A values of n3 (stride M): DFT-A per column, sixteen
//              columns interleaved in LDS (lanes run along m: 128-byte segments in and out),
//              times W_N^(m k3) = t1T[tau][m] * t2[m][q]  (k3 = tau + (A/16) q)   -> scratch[k3][m]
//   row pass   tile = 16 adjacent k3 rows x M (each row contiguous): DFT-M per row, fftshift
//              (km + M/2), log epilogue, then a 16 x M transpose through LDS so that the stores
//              X[A km + k3] run along k3 (64-byte segments)
// 28 B/sample of traffic (8 in, 8+8 scratch, 4 out) instead of the 44 of the three-pass form this
// replaces for N > 65536; the scratch is processed in chunks that stay in the Infinity Cache.
#include "fft_lds_core.h"

namespace sdrk {

// W = tile width in columns (16, or 8 for A = 1024 so that two 512-thread workgroups fit a CU)
template <int LOG2A, bool HAS_WINDOW, int W>
__global__ __launch_bounds__(LdsCfg<LOG2A>::T * W, (LdsCfg<LOG2A>::T * W >= 512 ? 4 : 3)) void col_pass_kernel(
    const float2* __restrict__ iq, size_t frame_stride, float2* __restrict__ scratch, size_t n_frames, int M,
    const float* __restrict__ window, const float2* __restrict__ twA, const float2* __restrict__ t1T,
    const float2* __restrict__ t2) {
    using C = LdsCfg<LOG2A>;
    constexpr int A = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0;
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];  // 17/16 A W elements: W interleaved columns
    const int tid = threadIdx.x;
    const int fr = tid & (W - 1), tau = tid / W;
    LdsTw<LOG2A> tw;
    lds_tw_init<LOG2A>(tw, twA, tau);
    const size_t nfft = (size_t)A * M;
    const int tiles = M / W;
    const size_t items = n_frames * (size_t)tiles;

    // With the grid a multiple of the tiles per frame every workgroup keeps the same tile position for all
    // its frames, so its 16 window coefficients per thread are loop invariants: keep them in registers
    // (saves the 4 B/sample of L2 traffic the window costs; not at 1024 threads, where VGPRs are capped at 128).
    constexpr bool WIN_REGS = HAS_WINDOW && (C::T * W <= 512);
    const bool fixed_tile = WIN_REGS && (gridDim.x % tiles) == 0 && W >= 16;
    float wreg[16];
    if (WIN_REGS && fixed_tile) {
        const int m_fixed = (int)(blockIdx.x % tiles) * W + fr;
#pragma unroll
        for (int q = 0; q < 16; ++q) wreg[q] = window[(size_t)(tau + T * q) * M + m_fixed];
    }

    for (size_t g = blockIdx.x; g < items; g += gridDim.x) {
        // W == 8: the two tiles that share each 128-byte line go to blocks b and b+8 (same XCD under the
        // round-robin placement; a speed hint only)
        size_t it = g;
        if (W == 8 && (items & 15) == 0) it = (g & ~(size_t)15) + ((g & 7) << 1) + ((g >> 3) & 1);
        const size_t f = it / tiles;
        const int m = (int)(it - f * tiles) * W + fr;
        // buffer addressing: wave-uniform descriptor on the frame, one 32-bit lane offset, uniform row steps
        const __amdgpu_buffer_rsrc_t rx = frame_rsrc(iq + f * frame_stride, (unsigned)(nfft * 8));
        const __amdgpu_buffer_rsrc_t rw = frame_rsrc(window, HAS_WINDOW ? (unsigned)(nfft * 4) : 0u);
        const int e0 = tau * M + m;          // element (n3 = tau, m)
        const int estep = T * M;             // n3 += T
        cf v[16];
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int q = i + C0 * j;
                v2f t = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, e0 * 8, q * estep * 8, 2));
                if (HAS_WINDOW) {
                    float w;
                    if (WIN_REGS && fixed_tile) w = wreg[q];
                    else w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, e0 * 4, q * estep * 4, 0));
                    t.x *= w;
                    t.y *= w;
                }
                v[i * R0 + j] = cf{t.x, t.y};
            }
        lds_fft_core<LOG2A, W>(v, lds_all, fr, tau, tw);
        // B[k3 = tau + T q] * W_N^(m k3),  W_N^(m k3) = W_N^(m tau) * W_N^(m T q)
        const __amdgpu_buffer_rsrc_t ro = frame_rsrc(scratch + f * nfft, (unsigned)(nfft * 8));
        const float2 bw = t1T[e0];
        const cf base = cf{bw.x, bw.y};
        const float4* __restrict__ row = reinterpret_cast<const float4*>(t2 + (size_t)m * 16);
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) {
            const float4 w = row[q2];
            const cf z0 = cmul(v[rev16(2 * q2)], cmul(base, cf{w.x, w.y}));
            const cf z1 = cmul(v[rev16(2 * q2 + 1)], cmul(base, cf{w.z, w.w}));
            const v2f s0 = {z0.x, z0.y}, s1 = {z1.x, z1.y};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, s0), ro, e0 * 8, (2 * q2) * estep * 8, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, s1), ro, e0 * 8, (2 * q2 + 1) * estep * 8, 0);
        }
    }
}

#ifndef SDRK_COL_W1024
#define SDRK_COL_W1024 16   // 8 (two 512-thread workgroups per CU, 64-byte segments) measured 10 % slower at N = 2^20
#endif
#ifndef SDRK_COL_W256
#define SDRK_COL_W256 16
#endif
#define COL_TILE_W(LOG2A) ((LOG2A) == 11 ? 8 : ((LOG2A) == 10 ? SDRK_COL_W1024 : ((LOG2A) == 8 ? SDRK_COL_W256 : 16)))
#define ROW_TILE_R(LOG2M) ((LOG2M) == 11 ? 8 : 16)

// ROWS = rows per tile (16, or 8 for M = 2048 so that the tile fits the LDS)
template <int LOG2M, int EPILOGUE, int ROWS>
__global__ __launch_bounds__(LdsCfg<LOG2M>::T * ROWS, (LdsCfg<LOG2M>::T * ROWS >= 512 ? 4 : 3)) void row_pass_kernel(
    const float2* __restrict__ scratch, void* __restrict__ out_raw, size_t n_frames, int A,
    const float2* __restrict__ twM, float eps, int shift) {
    using C = LdsCfg<LOG2M>;
    constexpr int M = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0;
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];  // 16 rows x SLOT, reused for the transpose
    const int tid = threadIdx.x;
    const int fr = tid / T, tau = tid - fr * T;
    float2* __restrict__ lds = lds_all + (size_t)fr * C::SLOT;
    LdsTw<LOG2M> tw;
    lds_tw_init<LOG2M>(tw, twM, tau);
    const size_t nfft = (size_t)A * M;
    const int tiles = A / ROWS;
    const size_t items = n_frames * (size_t)tiles;
    const int xor_q = shift ? 8 : 0;
    constexpr int WGT = T * ROWS;   // threads

    for (size_t it = blockIdx.x; it < items; it += gridDim.x) {
        const size_t f = it / tiles;
        const int k3_0 = (int)(it - f * tiles) * ROWS;
        const __amdgpu_buffer_rsrc_t ri = frame_rsrc(scratch + f * nfft + (size_t)k3_0 * M, (unsigned)(ROWS * M * 8));
        const int e0 = fr * M + tau;
        cf v[16];
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const v2f t = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, e0 * 8, (i + C0 * j) * T * 8, 0));
                v[i * R0 + j] = cf{t.x, t.y};
            }
        lds_fft_core<LOG2M, 1>(v, lds, 0, tau, tw);
        __syncthreads();  // all rows are through their last LDS reads: the buffer becomes the transpose tile
        if (EPILOGUE == EPI_LOGPSD) {
            float* __restrict__ tile = reinterpret_cast<float*>(lds_all);  // [km][ROWS + 1]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[rev16(q)];
                tile[(tau + T * (q ^ xor_q)) * (ROWS + 1) + fr] = logpsd_db(z.x, z.y, eps);
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t ro = frame_rsrc(static_cast<float*>(out_raw) + f * nfft + k3_0,
                                                         (unsigned)((nfft - k3_0) * 4));
            // element e = tid + WGT i of the ROWS x M tile: row r = e % ROWS (lanes), km = e / ROWS = tid / ROWS + T i
            const int r = tid & (ROWS - 1), km0 = tid / ROWS;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float val = tile[(km0 + T * i) * (ROWS + 1) + r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, (km0 * A + r) * 4,
                                                      i * T * A * 4, 2);
            }
        } else {
            float2* __restrict__ tile = lds_all;  // [km][ROWS + 1]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[rev16(q)];
                tile[(tau + T * (q ^ xor_q)) * (ROWS + 1) + fr] = make_float2(z.x, z.y);
            }
            __syncthreads();
            float2* __restrict__ o = static_cast<float2*>(out_raw) + f * nfft + k3_0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = tid + WGT * i;
                const int r = e & (ROWS - 1), km = e / ROWS;
                o[(size_t)km * A + r] = tile[km * (ROWS + 1) + r];
            }
        }
        __syncthreads();  // tile reads done before the next item's exchanges
    }
}

bool fft_tiled2_split(int nfft, int* log2a, int* log2m) {
    int lg = 0;
    while ((1 << lg) < nfft) ++lg;
    if ((1 << lg) != nfft || lg < 15 || lg > 22) return false;
    const int la = lg / 2 < 7 ? 7 : lg / 2;   // 2^15 -> 128 x 256 ; 2^17 -> 256 x 512 ; 2^20 -> 1024 x 1024 ; 2^22 -> 2048 x 2048
    *log2a = la;
    *log2m = lg - la;
    return true;
}

template <int LOG2A>
static hipError_t launch_col(const LaunchArgs& a, const float2* src, size_t nf, int M, unsigned grid_cap) {
    using C = LdsCfg<LOG2A>;
    constexpr int W = COL_TILE_W(LOG2A);
    const size_t lds_bytes = (size_t)(C::SLOT) * W * sizeof(float2);
    const size_t items = nf * (size_t)(M / W);
    if (W == 8) grid_cap = (unsigned)a.num_cus * 2;
    unsigned grid = (unsigned)(items < grid_cap ? items : grid_cap);
    if (W == 8 && grid >= 16) grid &= ~15u;
    const float2* twA = static_cast<const float2*>(a.d_twiddle_2p);
    const float2* t1T = twA + 2048 + 2048;
    const float2* t2 = t1T + (size_t)(C::T) * M;
    float2* scratch = static_cast<float2*>(a.d_scratch);
#define SDRK_COL(WIN)                                                                                            \
    do {                                                                                                         \
        auto kern = col_pass_kernel<LOG2A, WIN, W>;                                                              \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                     \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);              \
        if (e0 != hipSuccess) return e0;                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::T * W), lds_bytes, a.stream, src, a.frame_stride, scratch, nf, \
                           M, a.d_window, twA, t1T, t2);                                                            \
    } while (0)
    if (a.d_window) SDRK_COL(true); else SDRK_COL(false);
#undef SDRK_COL
    return hipGetLastError();
}

template <int LOG2M>
static hipError_t launch_row(const LaunchArgs& a, void* dst, size_t nf, int A, unsigned grid_cap) {
    using C = LdsCfg<LOG2M>;
    constexpr int ROWS = ROW_TILE_R(LOG2M);
    // LDS: the exchange area (ROWS x 17/16 M complex) or the complex transpose tile (M x (ROWS+1)), whichever is larger
    const size_t xch = (size_t)ROWS * C::SLOT, tile = (size_t)C::N * (ROWS + 1);
    const size_t lds_bytes = (xch > tile ? xch : tile) * sizeof(float2);
    const size_t items = nf * (size_t)(A / ROWS);
    const unsigned grid = (unsigned)(items < grid_cap ? items : grid_cap);
    const float2* twM = static_cast<const float2*>(a.d_twiddle_2p) + 2048;
    const float2* scratch = static_cast<const float2*>(a.d_scratch);
#define SDRK_ROW(E)                                                                                              \
    do {                                                                                                         \
        auto kern = row_pass_kernel<LOG2M, E, ROWS>;                                                                   \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                     \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);              \
        if (e0 != hipSuccess) return e0;                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::T * ROWS), lds_bytes, a.stream, scratch, dst, nf, A, twM, a.eps, \
                           a.shift);                                                                             \
    } while (0)
    if (a.epilogue == EPI_LOGPSD) SDRK_ROW(EPI_LOGPSD); else SDRK_ROW(EPI_COMPLEX);
#undef SDRK_ROW
    return hipGetLastError();
}

hipError_t launch_fft_tiled2(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    int la = 0, lm = 0;
    if (!fft_tiled2_split(a.nfft, &la, &lm)) return hipErrorInvalidValue;
    const int A = 1 << la, M = 1 << lm;
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const size_t out_elem = a.epilogue == EPI_LOGPSD ? sizeof(float) : sizeof(float2);
    // workgroups per CU: LDS 136 B per point of A (or M); at most 2048 threads
    auto cap = [&](int L) {
        size_t per_cu = (160 * 1024) / ((size_t)136 * L);
        const size_t by_threads = 2048 / L;
        if (per_cu > by_threads) per_cu = by_threads;
        if (per_cu > 4) per_cu = 4;
        if (per_cu < 1) per_cu = 1;
        return (unsigned)(a.num_cus * per_cu);
    };
    for (size_t f0 = 0; f0 < a.n_frames; f0 += a.scratch_frames) {
        const size_t nf = (a.n_frames - f0 < a.scratch_frames) ? a.n_frames - f0 : a.scratch_frames;
        const float2* src = iq + f0 * a.frame_stride;
        void* dst = static_cast<char*>(a.d_out) + f0 * (size_t)a.nfft * out_elem;
        hipError_t e;
        switch (la) {
            case 7: e = launch_col<7>(a, src, nf, M, cap(A)); break;
            case 8: e = launch_col<8>(a, src, nf, M, cap(A)); break;
            case 9: e = launch_col<9>(a, src, nf, M, cap(A)); break;
            case 10: e = launch_col<10>(a, src, nf, M, cap(A)); break;
            default: e = launch_col<11>(a, src, nf, M, cap(1024)); break;
        }
        if (e != hipSuccess) return e;
        switch (lm) {
            case 8: e = launch_row<8>(a, dst, nf, A, cap(M)); break;
            case 9: e = launch_row<9>(a, dst, nf, A, cap(M)); break;
            case 10: e = launch_row<10>(a, dst, nf, A, cap(M)); break;
            default: e = launch_row<11>(a, dst, nf, A, cap(1024)); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sdrk
