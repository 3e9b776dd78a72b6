// bluestein.hip — frame lengths that are not a power of two (the reference transforms whatever
// len(samples) is: np.fft.fft at app/sdr/streamer.py:119 with a non-default rx_buffer_size,
// streamer.py:10).  Chirp-z / Bluestein: with c[n] = exp(+i pi n^2 / N),
//   X[k] = conj(c[k]) * sum_n (x[n] w[n] conj(c[n])) * c[k-n]
// i.e. one length-M circular convolution (M = power of two >= 2N-1) done with the library's own
// power-of-two transforms:  conv = IFFT_M( FFT_M(a) * FFT_M(b) ),  IFFT(C) = conj(FFT(conj(C))) / M.
//   blu_pre   a[m] = x[m] w[m] conj(c[m]) (m < N), 0 (N <= m < M)
//   (FFT_M)   inner plan, complex epilogue
//   blu_mul   C'[m] = conj(A[m] * B[m])            B = FFT_M(b), b[m] = b[M-m] = c[m], precomputed
//   (FFT_M)
//   blu_post  X[k] = conj(c[k]) * conj(Y[k]) / M ; fftshift (index + N/2 mod N, odd N included) ;
//             20*log10(|X| + eps) or the complex value
// The chirp table is built on the host in double precision with n^2 reduced mod 2N.
#include "fft_lds_core.h"

namespace sdrk {

// L = M: whole zero-padded frames; L = N (compact): only the values that exist, at stride M
__global__ __launch_bounds__(256) void blu_pre_kernel(const float2* __restrict__ iq, size_t frame_stride,
                                                      size_t n_frames, int N, int M, int L, const float* __restrict__ window,
                                                      const float2* __restrict__ chirp, float2* __restrict__ a) {
    const size_t total = n_frames * (size_t)L;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += (size_t)gridDim.x * 256) {
        const size_t f = i0 / L;
        const int m = (int)(i0 - f * L);
        const size_t i = f * (size_t)M + m;
        float2 v = make_float2(0.f, 0.f);
        if (m < N) {
            float2 x = iq[f * frame_stride + m];
            if (window) { const float w = window[m]; x.x *= w; x.y *= w; }
            const float2 c = chirp[m];  // multiply by conj(c)
            v = make_float2(fmaf(x.x, c.x, x.y * c.y), fmaf(x.y, c.x, -(x.x * c.y)));
        }
        a[i] = v;
    }
}

__global__ __launch_bounds__(256) void blu_mul_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                      size_t n_frames, int M, float2* __restrict__ out) {
    const size_t total = n_frames * (size_t)M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float2 a = A[i], b = B[i % M];
        out[i] = make_float2(fmaf(a.x, b.x, -(a.y * b.y)), -fmaf(a.x, b.y, a.y * b.x));  // conj(a*b)
    }
}

template <int EPILOGUE>
__global__ __launch_bounds__(256) void blu_post_kernel(const float2* __restrict__ Y, const float2* __restrict__ chirp,
                                                       size_t n_frames, int N, int M, float eps, int shift,
                                                       void* __restrict__ out_raw) {
    const size_t total = n_frames * (size_t)N;
    const float inv_m = 1.0f / (float)M;
    const int rot = shift ? N / 2 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t f = i / N;
        const int k = (int)(i - f * N);
        const float2 y = Y[f * (size_t)M + k], c = chirp[k];
        // conj(c) * conj(y) = conj(c*y)
        const float re = fmaf(c.x, y.x, -(c.y * y.y)) * inv_m;
        const float im = -fmaf(c.x, y.y, c.y * y.x) * inv_m;
        int dst = k + rot;
        if (dst >= N) dst -= N;
        if (EPILOGUE == EPI_LOGPSD)
            static_cast<float*>(out_raw)[f * (size_t)N + dst] = logpsd_db(re, im, eps);
        else
            static_cast<float2*>(out_raw)[f * (size_t)N + dst] = make_float2(re, im);
    }
}

// ---- fused form for M <= 16384: the whole chirp-z of a frame in ONE kernel, one LDS residency.  The pre-multiply rides on
// the loads of the first transform; its result X[tau + T q] is in the thread's own registers in exactly the index pattern the
// next transform's input wants (x[tau + T (i + C0 j)] in slot i R0 + j), so the filter multiply (+ conjugation) and the hand-over
// to the second transform are register moves — no exchange, no trip through HBM —; the post-multiply / fftshift / log ride on the
// second transform's stores.  12 N bytes of HBM traffic per frame (the chirp and the filter spectrum come from L2) where the
// two-kernel form of round 3 moved 12 N + 16 M and the five-pass form 12 N + 64 M.
template <int LOG2M, int EPILOGUE>
__global__ __launch_bounds__(LdsCfg<LOG2M>::WG, LdsCfg<LOG2M>::WAVES) void blu_one_kernel(
    const float2* __restrict__ iq, size_t frame_stride, size_t n_frames, int N, const float* __restrict__ window,
    const float2* __restrict__ chirp, const float2* __restrict__ bspec, const float2* __restrict__ twM, float eps, int shift,
    void* __restrict__ out_raw) {
    using C = LdsCfg<LOG2M>;
    constexpr int M = C::N, P = C::P, R0 = C::R0, T = C::T, F = C::F, C0 = 16 / R0;
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];
    // the per-thread twiddle bases: constant over the frames, but at 1024 threads (128 registers) keeping them across the loop
    // costs spills — there they are fetched again per frame (L2 hits) from the opaque thread number below
    constexpr bool TW_PER_FRAME = C::WG >= 1024;
    LdsTw<LOG2M> tw;
    if (!TW_PER_FRAME) lds_tw_init<LOG2M>(tw, twM, (int)threadIdx.x % T);
    const size_t n_groups = (n_frames + F - 1) / F;
    const float inv_m = 1.0f / (float)M;
    const int rot = shift ? N / 2 : 0;
    for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        // (thread coordinates re-derived per group from an opaque copy of the thread number: hoisted out of this
        // persistent loop, the sixteen sample offsets and their `n < N` masks cost the registers the transform spills)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < C::WG);
        const int fr = tid / T, tau = tid - fr * T;
        if (TW_PER_FRAME) lds_tw_init<LOG2M>(tw, twM, tau);
        float2* __restrict__ lds = lds_all + (size_t)fr * C::SLOT;
        const size_t f = g * F + fr;
        const bool ok = f < n_frames;
        const float2* __restrict__ x = iq + (ok ? f : 0) * frame_stride;
        cf v[16];
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int n = tau + T * (i + C0 * j);
                cf a = cf{0.f, 0.f};
                if (ok && n < N) {
                    float2 s = x[n];
                    if (window) { const float w = window[n]; s.x *= w; s.y *= w; }
                    const float2 c = chirp[n];  // times conj(c)
                    a = cf{fmaf(s.x, c.x, s.y * c.y), fmaf(s.y, c.x, -(s.x * c.y))};
                }
                v[i * R0 + j] = a;
            }
        lds_fft_core<LOG2M, 1>(v, lds, 0, tau, tw);
        // A[k] at k = tau + T q (k = q for the one-pass lengths) sits in v[rev16(q)]: C'[k] = conj(A[k] B[k]) goes to the slot
        // that reads x[tau + T (i + C0 j)] with q = i + C0 j
        // (in place, four at a time: sixteen filter values fetched at once cost the 1024-thread lengths their registers)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = P == 1 ? q : tau + T * q;
            const cf a = v[rev16(q)];
            const float2 b = bspec[k];
            v[rev16(q)] = cf{fmaf(a.x, b.x, -(a.y * b.y)), -fmaf(a.x, b.y, a.y * b.x)};
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        cf u[16];
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) u[i * R0 + j] = v[rev16(i + C0 * j)];
        // (the second transform's exchange addresses are computed again from an opaque copy of the row index: kept from the first
        // transform they would be sixteen more live registers across it)
        int tau2 = tau;
        asm volatile("" : "+v"(tau2));
        __builtin_assume(tau2 >= 0 && tau2 < T);
        lds_fft_core<LOG2M, 1>(u, lds, 0, tau2, tw);
        if (ok) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = P == 1 ? q : tau2 + T * q;
                if (k < N) {
                    const cf y = u[rev16(q)];
                    const float2 c = chirp[k];
                    const float re = fmaf(c.x, y.x, -(c.y * y.y)) * inv_m;   // conj(c) * conj(y) = conj(c y)
                    const float im = -fmaf(c.x, y.y, c.y * y.x) * inv_m;
                    int dst = k + rot;
                    if (dst >= N) dst -= N;
                    if (EPILOGUE == EPI_LOGPSD)
                        static_cast<float*>(out_raw)[f * (size_t)N + dst] = logpsd_db(re, im, eps);
                    else
                        static_cast<float2*>(out_raw)[f * (size_t)N + dst] = make_float2(re, im);
                }
            }
        }
    }
}

template <int LOG2M>
static hipError_t blu_fused_n(const void* d_iq, size_t frame_stride, size_t n_frames, int N, const float* d_window,
                              const void* d_chirp, const void* d_bspec, const void* d_twM, float eps,
                              int shift, int epilogue, void* d_out, int num_cus, hipStream_t s) {
    using C = LdsCfg<LOG2M>;
    const size_t lds_bytes = (size_t)C::F * C::SLOT * sizeof(float2);
    size_t per_cu = (160 * 1024) / lds_bytes;
    if (per_cu > (size_t)(2048 / C::WG)) per_cu = 2048 / C::WG;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const size_t n_groups = (n_frames + C::F - 1) / C::F, cap = (size_t)num_cus * per_cu;
    const unsigned grid = (unsigned)(n_groups < cap ? n_groups : cap);
    auto k0 = blu_one_kernel<LOG2M, EPI_LOGPSD>;
    auto k1 = blu_one_kernel<LOG2M, EPI_COMPLEX>;
    static std::atomic<uint64_t> lds_set0{0}, lds_set1{0};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(epilogue == EPI_LOGPSD ? k0 : k1), lds_bytes,
                                      epilogue == EPI_LOGPSD ? lds_set0 : lds_set1);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(epilogue == EPI_LOGPSD ? k0 : k1, dim3(grid), dim3(C::WG), lds_bytes, s, static_cast<const float2*>(d_iq),
                       frame_stride, n_frames, N, d_window, static_cast<const float2*>(d_chirp),
                       static_cast<const float2*>(d_bspec), static_cast<const float2*>(d_twM), eps, shift, d_out);
    return hipGetLastError();
}

bool blu_fused_supports(int M) { return M >= 16 && M <= 16384; }

hipError_t launch_blu_fused(const void* d_iq, size_t frame_stride, size_t n_frames, int N, int M, const float* d_window,
                            const void* d_chirp, const void* d_bspec, const void* d_twM, float eps, int shift,
                            int epilogue, void* d_out, int num_cus, hipStream_t s) {
#define BLU_CASE(L) case (1 << L): return blu_fused_n<L>(d_iq, frame_stride, n_frames, N, d_window, d_chirp, d_bspec, d_twM, eps, shift, epilogue, d_out, num_cus, s);
    switch (M) {
        BLU_CASE(4) BLU_CASE(5) BLU_CASE(6) BLU_CASE(7) BLU_CASE(8) BLU_CASE(9) BLU_CASE(10) BLU_CASE(11) BLU_CASE(12)
        BLU_CASE(13) BLU_CASE(14)
        default: return hipErrorInvalidValue;
    }
#undef BLU_CASE
}

static unsigned blu_grid(size_t total, int num_cus) {
    size_t b = (total + 255) / 256, cap = (size_t)num_cus * 16;
    return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

hipError_t launch_blu_pre(const void* d_iq, size_t frame_stride, size_t n_frames, int N, int M, const float* d_window,
                          const void* d_chirp, void* d_a, int num_cus, hipStream_t s, bool compact) {
    const int L = compact ? N : M;
    hipLaunchKernelGGL(blu_pre_kernel, dim3(blu_grid(n_frames * (size_t)L, num_cus)), dim3(256), 0, s,
                       static_cast<const float2*>(d_iq), frame_stride, n_frames, N, M, L, d_window,
                       static_cast<const float2*>(d_chirp), static_cast<float2*>(d_a));
    return hipGetLastError();
}
hipError_t launch_blu_mul(const void* d_A, const void* d_B, size_t n_frames, int M, void* d_out, int num_cus,
                          hipStream_t s) {
    hipLaunchKernelGGL(blu_mul_kernel, dim3(blu_grid(n_frames * (size_t)M, num_cus)), dim3(256), 0, s,
                       static_cast<const float2*>(d_A), static_cast<const float2*>(d_B), n_frames, M,
                       static_cast<float2*>(d_out));
    return hipGetLastError();
}

hipError_t launch_blu_post(const void* d_Y, const void* d_chirp, size_t n_frames, int N, int M, float eps, int shift,
                           int epilogue, void* d_out, int num_cus, hipStream_t s) {
    dim3 g(blu_grid(n_frames * (size_t)N, num_cus)), b(256);
    if (epilogue == EPI_LOGPSD)
        hipLaunchKernelGGL((blu_post_kernel<EPI_LOGPSD>), g, b, 0, s, static_cast<const float2*>(d_Y),
                           static_cast<const float2*>(d_chirp), n_frames, N, M, eps, shift, d_out);
    else
        hipLaunchKernelGGL((blu_post_kernel<EPI_COMPLEX>), g, b, 0, s, static_cast<const float2*>(d_Y),
                           static_cast<const float2*>(d_chirp), n_frames, N, M, eps, shift, d_out);
    return hipGetLastError();
}

}  // namespace sdrk
