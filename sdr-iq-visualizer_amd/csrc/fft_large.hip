// fft_large.hip — frames longer than one workgroup's LDS: nfft = N1 * 4096,
// 2 <= N1 <= 1024 (nfft up to 2^22), the waterfall sizes of BASELINE.json
// configs 3 and 5 (N = 65536, N = 2^20).  Same boundary as the flagship kernel:
// window -> FFT -> fftshift -> 20*log10(|X|+eps)  (app/sdr/streamer.py:119,121).
//
// Two steps over a device scratch of complex64 (n = n1 + N1 n2, k = 4096 k1 + k2):
//   step A  for every n1: 4096-point FFT over n2 of x[n1 + N1 n2] (in-LDS core of
//           fft4096_core.h, strided loads), times W_N^(n1 k2)      -> B[n1][k2]
//   step B  for every k2: N1-point FFT over n1 (Stockham radix-2 in LDS on a
//           tile of adjacent k2 columns), log-PSD epilogue          -> X[4096 k1 + k2]
// The fftshift is a rotation of k1 by N1/2.  W_N^m (m = n1 k2 < N) is the
// product coarse[m >> 12] * fine[m & 4095] of two small tables; coarse is also
// the W_N1 table of step B.
#include "fft4096_core.h"

namespace sdrk {

// ---- step A ------------------------------------------------------------------
template <bool HAS_WINDOW>
__global__ __launch_bounds__(F4K_THREADS, 2) void fft_large_rows_kernel(
    const float2* __restrict__ iq, size_t frame_stride, float2* __restrict__ scratch,
    size_t n_frames, int n1_count, const float* __restrict__ window,
    const float2* __restrict__ tw4096, const float2* __restrict__ coarse,
    const float2* __restrict__ fine) {
    __shared__ float2 lds[F4K_XCH_ELEMS + F4K_TW_ELEMS];
    float2* __restrict__ tw256 = lds + F4K_XCH_ELEMS;
    float2* __restrict__ tw4k = tw256 + 256;
    const int tid = threadIdx.x;
    F4kAddr A = f4k_addr(tid);
    f4k_init_tables(tw256, tw4k, tw4096, tid, A);
    __syncthreads();

    // XCD-aware persistent schedule: blocks with equal blockIdx % 8 share an
    // L2, so give each such group one contiguous range of (frame, n1) items:
    // the N1 rows of a frame interleave inside the same 128-byte lines.
    const size_t items = n_frames * (size_t)n1_count;
    const unsigned groups = gridDim.x >= 8 ? 8 : 1;
    const unsigned grp = blockIdx.x % groups, slot = blockIdx.x / groups;
    const unsigned slots = (gridDim.x - grp + groups - 1) / groups;
    const size_t lo_item = items * grp / groups, hi_item = items * (grp + 1) / groups;

    for (size_t it = lo_item + slot; it < hi_item; it += slots) {
        const size_t f = it / n1_count;
        const int n1 = (int)(it - f * n1_count);
        const float2* __restrict__ x = iq + f * frame_stride;
        cf v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t n = (size_t)n1 + (size_t)n1_count * (tid + 256 * j);
            float2 t = x[n];
            if (HAS_WINDOW) {
                float w = window[n];
                t.x *= w;
                t.y *= w;
            }
            v[j] = cf{t.x, t.y};
        }
        f4k_transform(v, lds, tw256, tw4k, A, tid);
        float2* __restrict__ o = scratch + it * (size_t)F4K_N;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k2 = tid + 256 * j;
            const unsigned m = (unsigned)n1 * (unsigned)k2;
            float2 c = coarse[m >> 12], fn = fine[m & 4095];
            cf z = cmul(v[rev16(j)], cmul(cf{c.x, c.y}, cf{fn.x, fn.y}));
            o[k2] = make_float2(z.x, z.y);
        }
    }
}

// ---- step B ------------------------------------------------------------------
constexpr int FL_THREADS = 256;

template <int EPILOGUE>
__global__ __launch_bounds__(FL_THREADS) void fft_large_cols_kernel(
    const float2* __restrict__ scratch, void* __restrict__ out_raw, size_t n_frames, int n1_count,
    int log2n1, int tile, const float2* __restrict__ coarse /* W_N1^m */, float eps, int shift) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];  // 2 * n1_count * tile
    const int tid = threadIdx.x;
    const int tiles_per_frame = F4K_N / tile;
    const size_t items = n_frames * (size_t)tiles_per_frame;
    const int half = n1_count >> 1;
    const int rot = shift ? half : 0;
    const size_t nfft = (size_t)n1_count * F4K_N;
    const int elems = n1_count * tile;

    for (size_t it = blockIdx.x; it < items; it += gridDim.x) {
        const size_t f = it / tiles_per_frame;
        const int c0 = (int)(it - f * tiles_per_frame) * tile;
        float2* __restrict__ a = sm;
        float2* __restrict__ b = sm + elems;
        const float2* __restrict__ src = scratch + f * nfft + c0;
        for (int i = tid; i < elems; i += FL_THREADS) {
            const int n1 = i / tile, c = i - n1 * tile;
            a[i] = src[(size_t)n1 * F4K_N + c];
        }
        __syncthreads();
        for (int s = 0; s < log2n1; ++s) {
            const int Ns = 1 << s;
            for (int i = tid; i < half * tile; i += FL_THREADS) {
                const int p = i / tile, c = i - p * tile;
                const int k = p & (Ns - 1);
                float2 u0 = a[p * tile + c], u1 = a[(p + half) * tile + c];
                float2 w = coarse[k << (log2n1 - 1 - s)];
                float2 t1 = make_float2(fmaf(u1.x, w.x, -(u1.y * w.y)), fmaf(u1.x, w.y, u1.y * w.x));
                const int j = ((p - k) << 1) + k;
                b[j * tile + c] = make_float2(u0.x + t1.x, u0.y + t1.y);
                b[(j + Ns) * tile + c] = make_float2(u0.x - t1.x, u0.y - t1.y);
            }
            __syncthreads();
            float2* t = a;
            a = b;
            b = t;
        }
        // X[4096 k1 + c0 + c] -> output row ((k1 + rot) mod N1)
        if (EPILOGUE == EPI_LOGPSD) {
            float* __restrict__ o = static_cast<float*>(out_raw) + f * nfft + c0;
            for (int i = tid; i < elems; i += FL_THREADS) {
                const int k1 = i / tile, c = i - k1 * tile;
                float2 z = a[i];
                o[(size_t)((k1 + rot) & (n1_count - 1)) * F4K_N + c] = logpsd_db(z.x, z.y, eps);
            }
        } else {
            float2* __restrict__ o = static_cast<float2*>(out_raw) + f * nfft + c0;
            for (int i = tid; i < elems; i += FL_THREADS) {
                const int k1 = i / tile, c = i - k1 * tile;
                o[(size_t)((k1 + rot) & (n1_count - 1)) * F4K_N + c] = a[i];
            }
        }
        __syncthreads();
    }
}

hipError_t launch_fft_large(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    const int n1_count = a.nfft / F4K_N;
    int log2n1 = 0;
    while ((1 << log2n1) < n1_count) ++log2n1;
    int tile = F4K_N / n1_count;  // n1_count * tile <= 4096 complex per LDS half
    if (tile > 256) tile = 256;
    const size_t lds_b = (size_t)2 * n1_count * tile * sizeof(float2);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
    const float2* coarse = static_cast<const float2*>(a.d_twiddle_big);
    const float2* fine = coarse + 1024;
    float2* scratch = static_cast<float2*>(a.d_scratch);
    const size_t out_elem = a.epilogue == EPI_LOGPSD ? sizeof(float) : sizeof(float2);

    for (size_t f0 = 0; f0 < a.n_frames; f0 += a.scratch_frames) {
        const size_t nf = (a.n_frames - f0 < a.scratch_frames) ? a.n_frames - f0 : a.scratch_frames;
        // step A
        {
            size_t items = nf * (size_t)n1_count;
            size_t max_blocks = (size_t)a.num_cus * 2;
            unsigned grid = (unsigned)(items < max_blocks ? items : max_blocks);
            const float2* src = iq + f0 * a.frame_stride;
            if (a.d_window)
                hipLaunchKernelGGL((fft_large_rows_kernel<true>), dim3(grid), dim3(F4K_THREADS), 0, a.stream,
                                   src, a.frame_stride, scratch, nf, n1_count, a.d_window, tw, coarse, fine);
            else
                hipLaunchKernelGGL((fft_large_rows_kernel<false>), dim3(grid), dim3(F4K_THREADS), 0, a.stream,
                                   src, a.frame_stride, scratch, nf, n1_count, a.d_window, tw, coarse, fine);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        // step B
        {
            size_t items = nf * (size_t)(F4K_N / tile);
            size_t max_blocks = (size_t)a.num_cus * 2;
            unsigned grid = (unsigned)(items < max_blocks ? items : max_blocks);
            void* dst = static_cast<char*>(a.d_out) + f0 * (size_t)a.nfft * out_elem;
            if (a.epilogue == EPI_LOGPSD)
                hipLaunchKernelGGL((fft_large_cols_kernel<EPI_LOGPSD>), dim3(grid), dim3(FL_THREADS), lds_b,
                                   a.stream, scratch, dst, nf, n1_count, log2n1, tile, coarse, a.eps, a.shift);
            else
                hipLaunchKernelGGL((fft_large_cols_kernel<EPI_COMPLEX>), dim3(grid), dim3(FL_THREADS), lds_b,
                                   a.stream, scratch, dst, nf, n1_count, log2n1, tile, coarse, a.eps, a.shift);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace sdrk
