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
#include "cplx.h"
#include "kernels.h"

namespace sdrk {

__global__ __launch_bounds__(256) void blu_pre_kernel(const float2* __restrict__ iq, size_t frame_stride,
                                                      size_t n_frames, int N, int M, const float* __restrict__ window,
                                                      const float2* __restrict__ chirp, float2* __restrict__ a) {
    const size_t total = n_frames * (size_t)M;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t f = i / M;
        const int m = (int)(i - f * M);
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

static unsigned blu_grid(size_t total, int num_cus) {
    size_t b = (total + 255) / 256, cap = (size_t)num_cus * 16;
    return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

hipError_t launch_blu_pre(const void* d_iq, size_t frame_stride, size_t n_frames, int N, int M, const float* d_window,
                          const void* d_chirp, void* d_a, int num_cus, hipStream_t s) {
    hipLaunchKernelGGL(blu_pre_kernel, dim3(blu_grid(n_frames * (size_t)M, num_cus)), dim3(256), 0, s,
                       static_cast<const float2*>(d_iq), frame_stride, n_frames, N, M, d_window,
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
