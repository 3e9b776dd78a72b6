// fft4096_features.hip — the N = 4096 transform with the classifier's per-row measurements as its
// epilogue: window * x -> FFT -> fftshift -> 20*log10(|X| + eps)  (app/sdr/streamer.py:119,121), then,
// while the row is still on chip, the reductions of app/processing/classifier.py:163-212
// (row_features_core.h).  The row itself is written to HBM only if the caller wants it: a consumer
// that needs the ~20 scalars and the peak list per frame costs 8 B/sample of HBM traffic (the IQ read)
// instead of 8 + 4 (row written) + 4 (row read back by a separate reduction kernel).
//
// Same decomposition and LDS layouts as fft4096.hip, but a different balance: the reductions, not the HBM stream,
// bound this kernel (~8x the transform's instruction count, most of it latency-bound hand-offs between the four
// waves), so it is built for occupancy instead of for bytes in flight — no register prefetch of the next frame
// (32 VGPRs) and the window read from L2 instead of a 16 KiB LDS copy: <= 128 VGPRs and 37 KiB of LDS, FOUR
// workgroups per CU that fill each other's barrier waits (the flagship runs three).  After the last pass the
// 16 KiB row replaces the exchange buffer in LDS, followed by the reduction scratch.
// Complex arithmetic in the scalar form here (cplx.h): with four waves on a SIMD two plain float32 instructions of two waves
// issue in the four cycles one packed instruction takes, so packing buys no throughput, and this kernel is issue-bound, not
// chain-bound (measured: packed 19.1 ns per row, scalar 18.8; profiles/r05/packed_complex_ab.log).  Both forms do the same
// float operations in the same order: the rows equal the flagship's bit for bit.
#ifndef SDRK_F1_PACKED_CF
#define SDRK_F1_PACKED_CF 0
#endif
#undef SDRK_PACKED_CF
#define SDRK_PACKED_CF SDRK_F1_PACKED_CF
// With SDRK_PACKED_CF = 0 here, sdrk::cf / cmul / radix16 / f4k_transform are defined differently in this translation unit than
// in every other one.  That is sound only because device code is compiled and linked per file; with relocatable device code
// (-fgpu-rdc) or device LTO the two definitions would be merged silently.  (csrc/Makefile: neither flag, by design.)
#if defined(__CLANG_RDC__) && !SDRK_F1_PACKED_CF
#error "fft4096_features.hip redefines sdrk::cf for this file only: do not build it with -fgpu-rdc (one-definition rule)"
#endif
#include "fft4096_core.h"
#include "row_features_core.h"

namespace sdrk {

#ifndef F4KF_WAVES
#define F4KF_WAVES 4   // workgroups per CU = waves per SIMD
#endif

template <bool HAS_WINDOW>
__global__ __launch_bounds__(F4K_THREADS, F4KF_WAVES) void fft4096_features_kernel(
    const float2* __restrict__ iq, size_t frame_stride, float* __restrict__ out_db, size_t n_frames,
    const float* __restrict__ window, const float2* __restrict__ tw4096, float eps, int shift, RowFeatParams prm,
    double* __restrict__ stats, double* __restrict__ thr, int* __restrict__ idx, int* __restrict__ cnt) {
    // 16-byte aligned by declaration, not by being the kernel's only LDS object: RowFeatShared::bins is read and written
    // as 128-bit vectors
    __shared__ __attribute__((aligned(16))) float2 lds[F4K_XCH_ELEMS + F4K_TW_ELEMS];
    static_assert(F4K_XCH_ELEMS * sizeof(float2) >= F4K_N * sizeof(float) + sizeof(RowFeatShared) + 16,
                  "row + reduction scratch must fit the exchange buffer");
    static_assert((F4K_N * sizeof(float)) % alignof(RowFeatShared) == 0 && alignof(RowFeatShared) <= 16,
                  "RowFeatShared sits behind the row at an offset that keeps its alignment");
    float2* __restrict__ tw256 = lds + F4K_XCH_ELEMS;
    float2* __restrict__ tw1 = tw256 + 256;
    float* __restrict__ row = reinterpret_cast<float*>(lds);                                  // 4096 float32
    RowFeatShared& sh = *reinterpret_cast<RowFeatShared*>(reinterpret_cast<char*>(lds) + F4K_N * sizeof(float));

    {
        const int tid = threadIdx.x;
        F4kAddr A = f4k_addr(tid);
        f4k_init_tables(tw256, tw1, tw4096, tid, A);
    }
    __syncthreads();

    const int xor_k2 = shift ? 8 : 0;
    const __amdgpu_buffer_rsrc_t rwin = frame_rsrc(window, HAS_WINDOW ? F4K_N * 4 : 0);
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        // the thread's LDS / buffer offsets are re-derived per frame from an opaque copy of its number (a dozen
        // integer operations): hoisted out of this persistent loop they would be live across the reductions too
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < F4K_THREADS);
        const F4kAddr A = f4k_addr(tid);
        cf v[16];
        SDRK_PHASE("load_window");
        {
            const __amdgpu_buffer_rsrc_t r = frame_rsrc(iq + f * frame_stride, F4K_N * 8);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2f t = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, j * 2048, 2));
                v[j] = cf{t.x, t.y};
            }
        }
        float win[16];
        if (HAS_WINDOW) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                win[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rwin, tid * 4, j * 1024, 0));
        }
        SDRK_PHASE("transform");
        if (HAS_WINDOW) f4k_transform<true>(v, lds, tw256, tw1, A, tid, win);
        else f4k_transform(v, lds, tw256, tw1, A, tid);
        __syncthreads();   // every thread is through its last exchange read: the buffer becomes the row
        SDRK_PHASE("logpsd_row");
        // bin k = tid + 256 k2 -> index tid + 256 (k2 ^ xor)
        __amdgpu_buffer_rsrc_t w = frame_rsrc(out_db ? out_db + f * (size_t)F4K_N : nullptr, out_db ? F4K_N * 4 : 0);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const cf z = v[rev16(k2)];
            const float db = logpsd_db(z.x, z.y, eps);
            row[tid + 256 * (k2 ^ xor_k2)] = db;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, db), w, tid * 4, (k2 ^ xor_k2) * 1024, 2);
        }
        __syncthreads();
        row_features_wg(row, F4K_N, prm, sh, stats + f * 16, thr ? thr + f : nullptr,
                        idx ? idx + f * (size_t)prm.max_peaks : nullptr, cnt ? cnt + f : nullptr);
        SDRK_PHASE("loop_end");
        __syncthreads();   // row and scratch are overwritten by the next frame's first exchange
    }
}

hipError_t launch_fft4096_features(const LaunchArgs& a, int rank, float gamma, int min_distance, int max_peaks,
                                   double* d_stats, double* d_thr, int* d_idx, int* d_cnt) {
    if (a.n_frames == 0) return hipSuccess;
    if (a.nfft != F4K_N) return hipErrorInvalidValue;
    RowFeatParams prm{rank, gamma, min_distance, max_peaks};
    const size_t max_blocks = (size_t)a.num_cus * F4KF_WAVES;
    const unsigned grid = (unsigned)(a.n_frames < max_blocks ? a.n_frames : max_blocks);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
    float* out = static_cast<float*>(a.d_out);
    if (a.d_window)
        hipLaunchKernelGGL((fft4096_features_kernel<true>), dim3(grid), dim3(F4K_THREADS), 0, a.stream, iq, a.frame_stride,
                           out, a.n_frames, a.d_window, tw, a.eps, a.shift, prm, d_stats, d_thr, d_idx, d_cnt);
    else
        hipLaunchKernelGGL((fft4096_features_kernel<false>), dim3(grid), dim3(F4K_THREADS), 0, a.stream, iq, a.frame_stride,
                           out, a.n_frames, a.d_window, tw, a.eps, a.shift, prm, d_stats, d_thr, d_idx, d_cnt);
    return hipGetLastError();
}

}  // namespace sdrk
