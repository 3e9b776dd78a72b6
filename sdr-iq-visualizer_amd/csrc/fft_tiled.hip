// fft_tiled.hip — frames of N = 256 * R * 256 samples, R in {1,2,4,8,16}
// (N = 65536 … 2^20: the waterfall sizes of BASELINE.json configs 3 and 5), as two or
// three fully tiled passes.  Same boundary as the flagship kernel: window -> FFT ->
// fftshift -> 20*log10(|X|+eps)   (app/sdr/streamer.py:119,121).
//
// Index split:  n = m + M n3,  m = n1 + 256 n2 in [0, M),  M = 256 R
//               k = 256 km + k3,  km = k2 + R k1
//   X[k] = sum_m W_M^(m km) * W_N^(m k3) * [ sum_n3 x[m + M n3] W_256^(n3 k3) ]
//   and, for the M-point row transform over m (when R > 1):
//   W_M^(m km) = W_M^(n1 k2) * W_256^(n1 k1) * W_R^(n2 k2)
//
//   K1 col256   tile = 16 consecutive m  x 256 n3 (stride M): DFT-256 over n3, times
//               W_N^(m k3)  -> scratch[k3][m]           (128-byte segments in and out)
//   K2 mid<R>   one thread per (k3, n1): DFT-R over n2 in registers, times W_M^(n1 k2),
//               in place   -> scratch[k3][k2][n1]       (512 B per wave instruction)
//   K3 row256   tile = 16 consecutive k3 x 256 n1 (fixed k2): DFT-256 over n1,
//               fftshift (k1 ^ 128), log-PSD -> X[256 (k2 + R k1) + k3]
//                                                        (2 KiB rows in, 64-byte segments out)
// Every 256-point transform is two in-register radix-16 passes around one LDS exchange
// (16 points per thread, 256 threads per tile), the structure of fft4096_core.h.
// The complex64 scratch is sized to stay in the 256 MiB Infinity Cache, so only the
// algorithmic 8 B/sample in and 4 B/sample out have to reach HBM.
#include "fft4096_core.h"

namespace sdrk {

constexpr int FT_THREADS = 256;

__device__ __forceinline__ void radix8(cf (&v)[16]) {
    // DFT-8 of v[0..7] in place, natural output order.
    constexpr float R2 = 0.70710678118654752440f;
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    bfly4(e0, e1, e2, e3);
    bfly4(o0, o1, o2, o3);
    cf t1 = cf{(o1.x + o1.y) * R2, (o1.y - o1.x) * R2};   // * W8^1
    cf t2 = mul_mi(o2);                                    // * W8^2
    cf t3 = cf{(o3.y - o3.x) * R2, -(o3.x + o3.y) * R2};  // * W8^3
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + t1; v[5] = e1 - t1;
    v[2] = e2 + t2; v[6] = e2 - t2;
    v[3] = e3 + t3; v[7] = e3 - t3;
}

// ---- K1: DFT-256 down the stride-M columns of a 16-wide tile -------------------------
template <bool HAS_WINDOW>
__global__ __launch_bounds__(FT_THREADS, 3) void col256_kernel(
    const float2* __restrict__ iq, size_t frame_stride, float2* __restrict__ scratch, size_t n_frames,
    int M, const float* __restrict__ window, const float2* __restrict__ tw4096,
    const float2* __restrict__ t1 /* [m][p] = W_N^(m p) */, const float2* __restrict__ t2 /* [m][q] = W_N^(16 m q) */) {
    __shared__ float2 lds[4096 + 256];
    float2* __restrict__ tw256 = lds + 4096;  // [k][n] = W256^(n k)
    const int tid = threadIdx.x;
    const int c = tid & 15, hi = tid >> 4;
    tw256[tid] = tw4096[(16 * c * hi) & 4095];
    __syncthreads();
    const int b = hi & 1;
    const int x1r_even = c + 256 * hi + 16 * b, x1r_odd = c + 256 * hi - 16 * b;
    const size_t nfft = (size_t)M * 256;
    const int tiles = M / 16;
    const size_t items = n_frames * (size_t)tiles;

    for (size_t it = blockIdx.x; it < items; it += gridDim.x) {
        const size_t f = it / tiles;
        const int m = (int)(it - f * tiles) * 16 + c;
        const float2* __restrict__ x = iq + f * frame_stride + m;
        cf v[16];
        // thread (c, a=hi): n3 = a + 16 j
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t off = (size_t)(hi + 16 * j) * M;
            float2 t = x[off];
            if (HAS_WINDOW) {
                float w = window[off + m];
                t.x *= w;
                t.y *= w;
            }
            v[j] = cf{t.x, t.y};
        }
        radix16(v);  // -> p ; times W256^(a p)
#pragma unroll
        for (int p = 1; p < 16; ++p) {
            float2 w = tw256[16 * p + hi];
            v[rev16(p)] = cmul(v[rev16(p)], cf{w.x, w.y});
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p)
            lds[((p & 1) ? (tid ^ 16) : tid) + 256 * p] = make_float2(v[rev16(p)].x, v[rev16(p)].y);
        __syncthreads();
        // thread (c, p=hi): all a
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            float2 t = lds[((a & 1) ? x1r_odd : x1r_even) + 16 * a];
            v[a] = cf{t.x, t.y};
        }
        radix16(v);  // -> q ; k3 = p + 16 q
        // W_N^(m k3) = W_N^(m p) * W_N^(16 m q): one entry of t1 and the 128-byte row m of t2
        // (16 lanes with consecutive m read 2 KiB contiguous; the four p of a wave broadcast).
        float2* __restrict__ o = scratch + f * nfft + m;
        const float2 bw = t1[m * 16 + hi];
        const cf base = cf{bw.x, bw.y};
        const float4* __restrict__ row = reinterpret_cast<const float4*>(t2 + (size_t)m * 16);
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) {
            const float4 w = row[q2];
            cf z0 = cmul(v[rev16(2 * q2)], cmul(base, cf{w.x, w.y}));
            cf z1 = cmul(v[rev16(2 * q2 + 1)], cmul(base, cf{w.z, w.w}));
            o[(size_t)(hi + 32 * q2) * M] = make_float2(z0.x, z0.y);
            o[(size_t)(hi + 32 * q2 + 16) * M] = make_float2(z1.x, z1.y);
        }
    }
}

// ---- K2: DFT-R over n2 (stride 256) in registers, in place --------------------------------
template <int R>
__global__ __launch_bounds__(256) void mid_kernel(float2* __restrict__ scratch, size_t n_cols /* frames*256*256 */,
                                                  const float2* __restrict__ tw4096) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_cols; i += (size_t)gridDim.x * 256) {
        const int n1 = (int)(i & 255);
        float2* __restrict__ col = scratch + (i >> 8) * (size_t)(256 * R) + n1;
        cf v[16];
#pragma unroll
        for (int n2 = 0; n2 < R; ++n2) {
            float2 t = col[256 * n2];
            v[n2] = cf{t.x, t.y};
        }
        if (R == 2) bfly2(v[0], v[1]);
        if (R == 4) {
            bfly4(v[0], v[1], v[2], v[3]);
        }
        if (R == 8) radix8(v);
        if (R == 16) radix16(v);
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2) {
            const int slot = (R == 16) ? rev16(k2) : k2;
            cf z = v[slot];
            if (k2) {
                float2 w = tw4096[n1 * k2 * (16 / R)];  // W_M^(n1 k2), M = 256 R
                z = cmul(z, cf{w.x, w.y});
            }
            col[256 * k2] = make_float2(z.x, z.y);
        }
    }
}

// ---- K3: DFT-256 along 16 adjacent rows, transposed store + epilogue ------------------------
template <int EPILOGUE>
__global__ __launch_bounds__(FT_THREADS, 3) void row256_kernel(
    const float2* __restrict__ scratch, void* __restrict__ out_raw, size_t n_frames, int R,
    const float2* __restrict__ tw4096, float eps, int shift) {
    __shared__ float2 lds[16 * 272 + 256];
    float2* __restrict__ tw256 = lds + 16 * 272;
    const int tid = threadIdx.x;
    const int lo = tid & 15, hi = tid >> 4;
    tw256[tid] = tw4096[(16 * lo * hi) & 4095];
    __syncthreads();
    const size_t nfft = (size_t)R * 65536;
    const size_t per_frame = (size_t)R * 16;  // (k2, k3-tile) pairs
    const size_t items = n_frames * per_frame;
    const int xor_q = shift ? 8 : 0;  // k1 ^ 128  <=>  q ^ 8   (k1 = p + 16 q)

    for (size_t g = blockIdx.x; g < items; g += gridDim.x) {
        // keep the two tiles that share each 128-byte output line on one XCD (b and b+8)
        size_t it = g;
        if ((items & 15) == 0) it = (g & ~(size_t)15) + ((g & 7) << 1) + ((g >> 3) & 1);
        const size_t f = it / per_frame;
        const int rem = (int)(it - f * per_frame);
        const int k2 = rem >> 4, k3_0 = (rem & 15) * 16;
        // thread (u=lo, r=hi): row k3_0 + r, n1 = u + 16 j
        const float2* __restrict__ in = scratch + f * nfft + ((size_t)(k3_0 + hi) * R + k2) * 256 + lo;
        cf v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float2 t = in[16 * j];
            v[j] = cf{t.x, t.y};
        }
        radix16(v);  // -> p ; times W256^(u p)
#pragma unroll
        for (int p = 1; p < 16; ++p) {
            float2 w = tw256[16 * p + lo];
            v[rev16(p)] = cmul(v[rev16(p)], cf{w.x, w.y});
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p)
            lds[lo + 17 * hi + 272 * p] = make_float2(v[rev16(p)].x, v[rev16(p)].y);
        __syncthreads();
        // thread (r=lo, p=hi): all u
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float2 t = lds[u + 17 * lo + 272 * hi];
            v[u] = cf{t.x, t.y};
        }
        radix16(v);  // -> q ; k1 = p + 16 q
        if (EPILOGUE == EPI_LOGPSD) {
            float* __restrict__ o = static_cast<float*>(out_raw) + f * nfft + k3_0 + lo;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k1 = hi + 16 * (q ^ xor_q);
                cf z = v[rev16(q)];
                o[256 * ((size_t)k2 + (size_t)R * k1)] = logpsd_db(z.x, z.y, eps);
            }
        } else {
            float2* __restrict__ o = static_cast<float2*>(out_raw) + f * nfft + k3_0 + lo;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k1 = hi + 16 * (q ^ xor_q);
                cf z = v[rev16(q)];
                o[256 * ((size_t)k2 + (size_t)R * k1)] = make_float2(z.x, z.y);
            }
        }
    }
}

bool fft_tiled_supports(int nfft) {
    return nfft == 65536 || nfft == (1 << 17) || nfft == (1 << 18) || nfft == (1 << 19) || nfft == (1 << 20);
}

hipError_t launch_fft_tiled(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    const int R = a.nfft / 65536, M = 256 * R;
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
    const float2* t1 = static_cast<const float2*>(a.d_twiddle_big) + 1024 + 4096;  // after coarse, fine
    const float2* t2 = t1 + (size_t)M * 16;
    float2* scratch = static_cast<float2*>(a.d_scratch);
    const size_t out_elem = a.epilogue == EPI_LOGPSD ? sizeof(float) : sizeof(float2);
    const size_t max_blocks = (size_t)a.num_cus * 3;

    for (size_t f0 = 0; f0 < a.n_frames; f0 += a.scratch_frames) {
        const size_t nf = (a.n_frames - f0 < a.scratch_frames) ? a.n_frames - f0 : a.scratch_frames;
        {
            size_t items = nf * (size_t)(M / 16);
            unsigned grid = (unsigned)(items < max_blocks ? items : max_blocks);
            const float2* src = iq + f0 * a.frame_stride;
            if (a.d_window)
                hipLaunchKernelGGL((col256_kernel<true>), dim3(grid), dim3(FT_THREADS), 0, a.stream, src,
                                   a.frame_stride, scratch, nf, M, a.d_window, tw, t1, t2);
            else
                hipLaunchKernelGGL((col256_kernel<false>), dim3(grid), dim3(FT_THREADS), 0, a.stream, src,
                                   a.frame_stride, scratch, nf, M, a.d_window, tw, t1, t2);
        }
        if (R > 1) {
            size_t cols = nf * 65536;
            size_t blocks = (cols + 255) / 256;
            if (blocks > (size_t)a.num_cus * 8) blocks = (size_t)a.num_cus * 8;
            dim3 g((unsigned)blocks), b(256);
            switch (R) {
                case 2: hipLaunchKernelGGL((mid_kernel<2>), g, b, 0, a.stream, scratch, cols, tw); break;
                case 4: hipLaunchKernelGGL((mid_kernel<4>), g, b, 0, a.stream, scratch, cols, tw); break;
                case 8: hipLaunchKernelGGL((mid_kernel<8>), g, b, 0, a.stream, scratch, cols, tw); break;
                default: hipLaunchKernelGGL((mid_kernel<16>), g, b, 0, a.stream, scratch, cols, tw); break;
            }
        }
        {
            size_t items = nf * (size_t)R * 16;
            unsigned grid = (unsigned)(items < max_blocks ? items : max_blocks);
            if (grid >= 16) grid &= ~15u;  // whole groups of 16 for the XCD pairing permutation
            void* dst = static_cast<char*>(a.d_out) + f0 * (size_t)a.nfft * out_elem;
            if (a.epilogue == EPI_LOGPSD)
                hipLaunchKernelGGL((row256_kernel<EPI_LOGPSD>), dim3(grid), dim3(FT_THREADS), 0, a.stream,
                                   scratch, dst, nf, R, tw, a.eps, a.shift);
            else
                hipLaunchKernelGGL((row256_kernel<EPI_COMPLEX>), dim3(grid), dim3(FT_THREADS), 0, a.stream,
                                   scratch, dst, nf, R, tw, a.eps, a.shift);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sdrk
