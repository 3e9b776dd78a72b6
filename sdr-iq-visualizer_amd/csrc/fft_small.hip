// fft_small.hip — general power-of-two frames, 2 <= nfft <= 4096, entirely in
// LDS: window -> Stockham radix-2 autosort FFT -> fftshift -> log-PSD (or the
// complex spectrum).  One workgroup per frame, grid-stride over frames.
//
// This is the catch-all behind the same boundary as the flagship kernel
// (reference call site app/sdr/streamer.py:119-121 with len(samples) != 4096,
// i.e. a non-default rx_buffer_size, streamer.py:10); nfft == 4096 normally
// goes to fft4096.hip and comes here only for A/B parity checks.
#include "cplx.h"
#include "kernels.h"

namespace sdrk {

constexpr int FS_THREADS = 256;

template <int EPILOGUE>
__global__ __launch_bounds__(FS_THREADS) void fft_small_kernel(
    const float2* __restrict__ iq, size_t frame_stride, void* __restrict__ out_raw,
    size_t n_frames, int nfft, int log2n, const float* __restrict__ window,
    const float2* __restrict__ tw /* W_nfft^m, m < nfft */, float eps, int shift) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];  // 2 * nfft
    const int tid = threadIdx.x;
    const int half = nfft >> 1;
    const int rot = shift ? half : 0;

    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        float2* __restrict__ a = sm;
        float2* __restrict__ b = sm + nfft;
        const float2* __restrict__ x = iq + f * frame_stride;
        for (int i = tid; i < nfft; i += FS_THREADS) {
            float2 t = x[i];
            if (window) {
                float w = window[i];
                t.x *= w;
                t.y *= w;
            }
            a[i] = t;
        }
        __syncthreads();
        // Stockham radix-2: stage s combines sub-transforms of length Ns = 2^s.
        for (int s = 0; s < log2n; ++s) {
            const int Ns = 1 << s;
            for (int i = tid; i < half; i += FS_THREADS) {
                const int k = i & (Ns - 1);
                float2 u0 = a[i], u1 = a[i + half];
                float2 w = tw[k << (log2n - 1 - s)];  // exp(-2 pi i k / (2 Ns))
                float2 t1 = make_float2(fmaf(u1.x, w.x, -(u1.y * w.y)), fmaf(u1.x, w.y, u1.y * w.x));
                const int j = ((i - k) << 1) + k;
                b[j] = make_float2(u0.x + t1.x, u0.y + t1.y);
                b[j + Ns] = make_float2(u0.x - t1.x, u0.y - t1.y);
            }
            __syncthreads();
            float2* t = a;
            a = b;
            b = t;
        }
        if (EPILOGUE == EPI_LOGPSD) {
            float* __restrict__ o = static_cast<float*>(out_raw) + f * (size_t)nfft;
            for (int k = tid; k < nfft; k += FS_THREADS) {
                float2 z = a[(k + rot) & (nfft - 1)];
                o[k] = logpsd_db(z.x, z.y, eps);
            }
        } else {
            float2* __restrict__ o = static_cast<float2*>(out_raw) + f * (size_t)nfft;
            for (int k = tid; k < nfft; k += FS_THREADS) o[k] = a[(k + rot) & (nfft - 1)];
        }
        __syncthreads();  // LDS is reused by the next frame
    }
}

hipError_t launch_fft_small(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    int log2n = 0;
    while ((1 << log2n) < a.nfft) ++log2n;
    size_t lds = (size_t)2 * a.nfft * sizeof(float2);
    size_t per_cu = 160 * 1024 / (lds > 8192 ? lds : 8192);
    if (per_cu > 8) per_cu = 8;
    size_t max_blocks = (size_t)a.num_cus * per_cu;
    unsigned grid = (unsigned)(a.n_frames < max_blocks ? a.n_frames : max_blocks);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
    if (a.epilogue == EPI_LOGPSD)
        hipLaunchKernelGGL((fft_small_kernel<EPI_LOGPSD>), dim3(grid), dim3(FS_THREADS), lds, a.stream,
                           iq, a.frame_stride, a.d_out, a.n_frames, a.nfft, log2n, a.d_window, tw,
                           a.eps, a.shift);
    else
        hipLaunchKernelGGL((fft_small_kernel<EPI_COMPLEX>), dim3(grid), dim3(FS_THREADS), lds, a.stream,
                           iq, a.frame_stride, a.d_out, a.n_frames, a.nfft, log2n, a.d_window, tw,
                           a.eps, a.shift);
    return hipGetLastError();
}

}  // namespace sdrk
