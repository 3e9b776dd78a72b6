// fft4096.hip — the flagship kernel: window * x -> 4096-point complex FFT ->
// fftshift -> 20*log10(|X| + eps), one HBM read (8 B/sample) and one HBM write
// (4 B/sample) per frame, everything in between in registers and LDS.
//
// Replaces, per frame, the reference's
//   app/sdr/streamer.py:119  fft_data = np.fft.fftshift(np.fft.fft(samples))
//   app/sdr/streamer.py:121  power_db = 20 * np.log10(np.abs(fft_data) + 1e-12)
//
// Decomposition (N = 4096 = 16*16*16, one 256-thread workgroup per frame, each
// thread owns 16 points; three in-register radix-16 passes, two LDS exchanges):
//   n = n0 + 16 n1 + 256 n2,   k = k0 + 16 k1 + 256 k2
//   pass 1  thread r=(n0,n1): DFT-16 over n2, times W4096^(r k0)   -> (n0,n1,k0)
//   pass 2  thread q=(n0,k0): DFT-16 over n1, times W256^(n0 k1)   -> (n0,k0,k1)
//   pass 3  thread p=(k0,k1): DFT-16 over n0                       -> X[p + 256 k2]
// Loads are x[r + 256 j]: every wave instruction reads 64 consecutive complex64
// (512 B); stores are out[p + 256 j]: 64 consecutive float32 (256 B).
// The fftshift of streamer.py:119 is a half rotation for even N: k2 ^= 8.
//
// LDS exchange layouts (complex64 units), every ds_read_b64 / ds_write_b64
// bank-conflict free on gfx950 (64 dword banks for b64 reads, 32 for writes) and
// every address of the form (one of two per-thread bases) + immediate:
//   exchange 1:  a1(n0,n1,k0) = n0 + 16 (n1 ^ (k0 & 1)) + 256 k0       (XOR swizzle)
//   exchange 2:  a2(n0,k0,k1) = k0 + 16 k1 + 257 n0                    (+1 padding)
//
// Global traffic goes through buffer instructions (wave-uniform descriptor on
// the frame, one VGPR byte offset, SGPR row offsets): no 64-bit address VGPRs.
// The kernel is persistent (grid-stride over frames) and software-pipelined:
// the next frame's 16 loads per thread are in flight while the current frame is
// transformed, so each resident workgroup keeps 32 KiB of HBM reads outstanding.
#include "fft4096_core.h"

#ifndef F4K_WINREG
#define F4K_WINREG 0   // 1: window coefficients in 16 VGPRs per thread instead of a 16 KiB LDS copy
#endif
#ifndef F4K_NT
#define F4K_NT 2       // cache-policy bits of the streaming loads/stores (2 = nt)
#endif

namespace sdrk {

// (Tried and dropped, A/B on the same buffers: two frames prefetched (2 % slower, again in round 5 at 140 VGPRs), a sqrt-free
// log epilogue (1.5 % slower), and sending the row through LDS once more so that it leaves as four 16-byte stores
// per thread instead of sixteen 4-byte ones (1.0-1.5 % slower: two more barriers per frame cost more than the
// narrower stores do); issuing the prefetch through inline asm with an exact `s_waitcnt vmcnt(16)` in front of its first
// use — hipcc waits vmcnt(0) there, i.e. also for the previous frame's stores — changed nothing either: with three
// workgroups per CU another wave always has work while one waits for its store acknowledgements.)
template <bool HAS_WINDOW, int EPILOGUE>
__global__ __launch_bounds__(F4K_THREADS, F4K_WAVES) void fft4096_kernel(
    const float2* __restrict__ iq, size_t frame_stride, void* __restrict__ out_raw,
    size_t n_frames, const float* __restrict__ window, const float2* __restrict__ tw4096,
    float eps, int shift) {
    __shared__ float2 lds[F4K_XCH_ELEMS + F4K_TW_ELEMS + ((HAS_WINDOW && !F4K_WINREG) ? F4K_N / 2 : 0)];
    float2* __restrict__ tw256 = lds + F4K_XCH_ELEMS;  // [k][n] = W256^(n k)
    float2* __restrict__ tw1 = tw256 + 256;            // W4096^tid
    float* __restrict__ lds_win = reinterpret_cast<float*>(tw1 + 256);

    const int tid = threadIdx.x;
    F4kAddr A = f4k_addr(tid);
    f4k_init_tables(tw256, tw1, tw4096, tid, A);
#if F4K_WINREG
    float win[16];
    if (HAS_WINDOW) {
#pragma unroll
        for (int j = 0; j < 16; ++j) win[j] = window[tid + 256 * j];
    }
#else
    if (HAS_WINDOW) {
#pragma unroll
        for (int j = 0; j < 16; ++j) lds_win[tid + 256 * j] = window[tid + 256 * j];
    }
#endif
    __syncthreads();

    const int xor_k2 = shift ? 8 : 0;
    const int voff_in = tid * 8;
    constexpr int OUT_ELEM = (EPILOGUE == EPI_LOGPSD ? 4 : 8);
    const int voff_out = tid * OUT_ELEM;

    const size_t first = blockIdx.x;
    const size_t step = gridDim.x;

    // Software pipeline: each workgroup keeps the loads of its next frame in flight while it transforms the
    // current one.  (Two frames ahead — 64 KiB per workgroup in flight — measured 2.5 % slower against the
    // same-buffers copy: 1.055x instead of 1.030x its time.)
    auto issue = [&](v2u (&x)[16], size_t fr) {
        if (fr >= n_frames) fr = first;  // harmless re-read past the end
        __amdgpu_buffer_rsrc_t r = frame_rsrc(iq + fr * frame_stride, F4K_N * 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(r, voff_in, j * 2048, F4K_NT);
    };
    auto process = [&](v2u (&x)[16], size_t f) {
        cf v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f t = __builtin_bit_cast(v2f, x[j]);
            v[j] = cf{t.x, t.y};
        }
        issue(x, f + step);
        if (HAS_WINDOW) {
#if !F4K_WINREG
            float win[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) win[j] = lds_win[tid + 256 * j];
#endif
            f4k_transform<true>(v, lds, tw256, tw1, A, tid, win);
        } else {
            f4k_transform(v, lds, tw256, tw1, A, tid);
        }
        // ---- epilogue + store: bin k = tid + 256 k2 -> index tid + 256 (k2 ^ xor) ----
        __amdgpu_buffer_rsrc_t w = frame_rsrc(
            static_cast<char*>(out_raw) + f * (size_t)(F4K_N * OUT_ELEM), F4K_N * OUT_ELEM);
        if (EPILOGUE == EPI_LOGPSD) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                cf z = v[rev16(k2)];
                float db = logpsd_db(z.x, z.y, eps);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, db), w, voff_out,
                                                      (k2 ^ xor_k2) * 1024, F4K_NT);
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                cf z = v[rev16(k2)];
                v2f o = {z.x, z.y};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), w, voff_out,
                                                      (k2 ^ xor_k2) * 2048, 0);
            }
        }
    };
    v2u nxt[16];
    issue(nxt, first);
    for (size_t f = first; f < n_frames; f += step) process(nxt, f);
}

hipError_t launch_fft4096(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    // Persistent grid: F4K_WAVES workgroups per CU.
    size_t max_blocks = (size_t)a.num_cus * F4K_WAVES;
    unsigned grid = (unsigned)(a.n_frames < max_blocks ? a.n_frames : max_blocks);
    dim3 g(grid), b(F4K_THREADS);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
#define SDRK_LAUNCH(W, E)                                                                            \
    hipLaunchKernelGGL((fft4096_kernel<W, E>), g, b, 0, a.stream, iq, a.frame_stride, a.d_out,           \
                       a.n_frames, a.d_window, tw, a.eps, a.shift)
    if (a.epilogue == EPI_LOGPSD) {
        if (a.d_window) SDRK_LAUNCH(true, EPI_LOGPSD); else SDRK_LAUNCH(false, EPI_LOGPSD);
    } else {
        if (a.d_window) SDRK_LAUNCH(true, EPI_COMPLEX); else SDRK_LAUNCH(false, EPI_COMPLEX);
    }
#undef SDRK_LAUNCH
    return hipGetLastError();
}

}  // namespace sdrk
