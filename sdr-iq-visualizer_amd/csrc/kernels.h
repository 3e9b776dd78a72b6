// kernels.h — internal interface between the C ABI (sdrk_api.hip) and the
// gfx950 kernels.  Nothing here is exported from the shared library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

// SDRK_PHASE("name"): a comment in the generated assembly that tools/phase_budget.py cuts a kernel's instruction stream at
// (builds with -DSDRK_PHASE_MARKS only: the memory clobber would otherwise fence the scheduler in the shipped code)
#ifdef SDRK_PHASE_MARKS
#define SDRK_PHASE(name) asm volatile("; SDRK_PHASE " name ::: "memory")
#else
#define SDRK_PHASE(name) do { } while (0)
#endif

namespace sdrk {

enum Epilogue : int {
    EPI_LOGPSD = 0,   // float32 20*log10(|X| + eps)          (streamer.py:121)
    EPI_COMPLEX = 1,  // complex64 X                           (streamer.py:119 only)
    // The chirp-z transform of large frames that are not a power of two (bluestein.hip) runs its two multiplies on the stores of
    // the inner transforms' row passes (fft_tiled2.hip) instead of as launches of their own — LaunchArgs::d_epi_tab, epi_n_out:
    EPI_BLU_MUL = 2,       // complex64 conj(X[k] * tab[k]), natural order (shift must be 0): the filter multiply + conjugation
    EPI_BLU_POST_LOG = 3,  // k < n_out only: y = conj(tab[k] X[k]) / nfft, stored at (k + n_out/2) mod n_out (or k), as dB,
    EPI_BLU_POST_C64 = 4,  // ... or as complex64; rows of n_out values (the transform's own length is nfft >= 2 n_out - 1)
};

struct LaunchArgs {
    const void* d_iq = nullptr;        // complex64, frame f at sample f*frame_stride
    size_t frame_stride = 0;           // samples between frame starts
    void* d_out = nullptr;             // float32[n_frames][nfft] or complex64[...]
    size_t n_frames = 0;
    int nfft = 0;
    const float* d_window = nullptr;   // nfft floats or nullptr (rectangular)
    const void* d_twiddle = nullptr;   // complex64 W_nfft^m, m in [0,nfft) (or [0,4096) for big plans)
    float eps = 1e-12f;
    int shift = 1;
    int epilogue = EPI_LOGPSD;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    void* d_scratch = nullptr;         // large-N plans: complex64 scratch, scratch_frames*nfft
    size_t scratch_frames = 0;
    const void* d_twiddle_2p = nullptr;   // two-pass tiled plans: W_A[2048] W_M[2048] t1T[(A/16)*M] t2[M*16]
    // two-pass tiled plans, overlapped form (fft_tiled2.hip): the row pass of chunk i runs on `stream2` beside the
    // col pass of chunk i+1 on `stream`, each on its own half of the scratch; col_cus / row_cus size the two
    // persistent grids so that both kernels are resident together (0: num_cus)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_col[2] = {nullptr, nullptr}, ev_row[2] = {nullptr, nullptr}, ev_fork = nullptr;
    int col_cus = 0, row_cus = 0;
    // optional second output of the frame lengths for which fft_tiled2_has_mip() holds: every row max-hold-decimated by 16,
    // float32[n_frames][nfft / 16] as [band = k3 / 16][km] (fft_tiled2.hip, row_pass_wave_kernel<…, MIP>); ignored elsewhere
    float* d_mip = nullptr;
    // out (host, optional): set by the launcher that honoured d_mip — whoever reads the companion rows later trusts THIS, not
    // its own idea of which plans write them (a new row-pass variant or build switch cannot then make the two disagree)
    bool* mip_written = nullptr;
    // EPI_BLU_*: the table the row pass multiplies by (filter spectrum: nfft entries; chirp: n_out entries) and the row length
    const float2* d_epi_tab = nullptr;
    int epi_n_out = 0;
    // two-pass plans: samples that exist per input frame (0 = nfft).  The col pass clips its loads to them — the rest of the frame
    // reads as zeros without touching memory (buffer bounds check): the zero padding of the chirp-z path's inner transforms
    size_t in_valid = 0;
};

#ifdef __HIPCC__   // device helpers (the host-only sanitizer build of sdrk_api.hip, tests/fake_hip, compiles this header with g++)
// 20*log10(sqrt(re^2+im^2) + eps), the expression order of streamer.py:121:
// |X| first, then the additive floor, then the log.  v_sqrt_f32 / v_log_f32 are
// 1-ulp approximations; the result is within ~2e-5 dB of numpy's float32 path
// over the float32 range (checked by tests/test_parity_gpu.py).
__device__ __forceinline__ float logpsd_db(float re, float im, float eps) {
    float p = fmaf(re, re, im * im);
    float mag = __builtin_amdgcn_sqrtf(p);
    return __builtin_amdgcn_logf(mag + eps) * 6.02059991327962390427f;  // log2 -> 20*log10
}

// Same value without the square root when the floor cannot change the float32
// sum: for |X| >= eps * 2^25, |X| + eps rounds to |X| (or its neighbour), so
// 20*log10(|X| + eps) = 10*log10(|X|^2) to within half an ulp of |X|.  `thresh`
// is (eps * 2^25)^2; the caller takes this path only when every lane of the
// wave is above it (wave-uniform branch), else logpsd_db().
__device__ __forceinline__ float logpsd_db_fast(float p) {
    return __builtin_amdgcn_logf(p) * 3.01029995663981195213f;  // log2 -> 10*log10
}

#endif  // __HIPCC__

// Kernels that need more than 64 KiB of dynamic LDS must opt in with hipFuncSetAttribute, which acts on
// the CURRENT device's function object: a process that drives several GPUs (sharding.py, channels.py)
// has to do it once per device.  `mask` is the per-kernel-instantiation record (bit d = done on device d).
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t lds_bytes, std::atomic<uint64_t>& mask) {
    if (lds_bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = (dev >= 0 && dev < 64) ? (uint64_t)1 << dev : 0;   // devices >= 64: set every time
    if (bit && (mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess && bit) mask.fetch_or(bit, std::memory_order_release);
    return e;
}

hipError_t launch_fft4096(const LaunchArgs& a);
hipError_t launch_fft_small(const LaunchArgs& a);   // 2 <= nfft <= 2048: Stockham radix-2 in LDS (N = 2, 4, 8 and far-apart frames)
bool fft_lds_supports(int nfft);                     // 16 .. 16384 except 4096: registers + LDS, one pass over HBM
hipError_t launch_fft_lds(const LaunchArgs& a);
bool fft_tiled2_split(int nfft, int* log2a, int* log2m);   // 2^15 .. 2^22: N = A * M, both in LDS
hipError_t launch_fft_tiled2(const LaunchArgs& a);
bool fft_tiled2_has_mip(int nfft, int epilogue);           // does the transform honour LaunchArgs::d_mip at this length?
hipError_t launch_synth_fill(uint32_t seed, uint64_t first_frame, size_t n_frames, int nfft,
                             void* d_iq, hipStream_t stream);

// N = 65536 in one persistent launch, intermediate ring resident in each XCD's L2 (fft_fused64k.hip)
size_t fused64k_ring_bytes();
size_t fused64k_ctrl_words();
unsigned fused64k_sets(int num_cus);      // sets of 32 workgroups a launch forms; ctrl[0] must equal it afterwards
hipError_t launch_fused64k(const LaunchArgs& a, void* d_ring, unsigned* d_ctrl);
// arbitrary frame lengths (bluestein.hip)
bool blu_fused_supports(int M);
hipError_t launch_blu_fused(const void* d_iq, size_t frame_stride, size_t n_frames, int N, int M, const float* d_window,
                            const void* d_chirp, const void* d_bspec, const void* d_twM, float eps, int shift,
                            int epilogue, void* d_out, int num_cus, hipStream_t s);
// compact: write only the N values of each frame (at stride M) and leave the padding to the reader's bounds check (in_valid)
hipError_t launch_blu_pre(const void* d_iq, size_t frame_stride, size_t n_frames, int N, int M, const float* d_window,
                          const void* d_chirp, void* d_a, int num_cus, hipStream_t s, bool compact = false);
hipError_t launch_blu_mul(const void* d_A, const void* d_B, size_t n_frames, int M, void* d_out, int num_cus,
                          hipStream_t s);
hipError_t launch_blu_post(const void* d_Y, const void* d_chirp, size_t n_frames, int N, int M, float eps, int shift,
                           int epilogue, void* d_out, int num_cus, hipStream_t s);
// per-row measurements (row_features.hip): stats[16], adaptive threshold, peak list; d_thr/d_idx/d_cnt may be null
hipError_t launch_row_features(const float* d_rows, size_t n_rows, int nfft, int rank, float gamma, int min_distance,
                               int max_peaks, double* d_stats, double* d_thr, int* d_idx, int* d_cnt, int num_cus,
                               hipStream_t s);
// the per-row arithmetic behind the scans, for a batch of packed results -> SDRK_FEAT_* planes (row_features.hip)
hipError_t launch_feature_finalize(const double* d_stats, const double* d_thr, const int* d_idx, const int* d_cnt,
                                   size_t n_rows, int nfft, float gamma, int max_peaks, const double* d_freqs,
                                   double* d_out, hipStream_t s);
// the N = 4096 transform with those measurements as its epilogue (fft4096_features.hip); a.d_out may be null
hipError_t launch_fft4096_features(const LaunchArgs& a, int rank, float gamma, int min_distance, int max_peaks,
                                   double* d_stats, double* d_thr, int* d_idx, int* d_cnt);
hipError_t launch_row_peaks(const float* d_rows, size_t n_rows, int nfft, const double* d_thr, int min_distance,
                            int max_peaks, int* d_idx, int* d_count, hipStream_t s);
hipError_t launch_decimate_rows(const float* d_ring, int nfft, int maxlen, int start_slot, int n_rows, int factor,
                                int mode, float* d_out, hipStream_t stream);
// the same max-hold read-out from the by-16 rows the transform left beside the ring (factor a multiple of 16)
hipError_t launch_decimate_mip(const float* d_mip_ring, int nfft, int maxlen, int start_slot, int n_rows, int factor,
                               float* d_out, hipStream_t stream);
hipError_t launch_stream_mix(const void* d_in, void* d_out, size_t n_frames4096, int num_cus, hipStream_t stream);
hipError_t launch_copy_1to1(const void* d_in, void* d_out, size_t bytes, int num_cus, int blocks_per_cu, hipStream_t stream);
hipError_t launch_power_mean(const void* d_spec, size_t n_frames, int nfft, float scale, float* d_out,
                             hipStream_t stream);

}  // namespace sdrk
