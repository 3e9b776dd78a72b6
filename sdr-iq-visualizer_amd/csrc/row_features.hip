// row_features.hip — per-row reductions over power_db rows, on the device, so that the immediate
// consumer of every spectrum row does not force a D2H of the row.
//
// These are the O(N) measurements behind the reference's rule-based classifier
// (app/processing/classifier.py): max and 20th-percentile noise floor (:45-46,179-181), occupied
// bandwidth at -3/-10/-20 dB (:163-170), spectral flatness (:183-189), kurtosis (:191-198), the
// adaptive threshold (:55) and the greedy peak list (:200-212).  The rule ladder (:69-122) and the
// 12-frame smoothing (:125-139) are scalar application logic and stay with the caller.
//
// row_features_kernel<STAGE, SHORT, NCONST>  one workgroup per row; the row is read from HBM ONCE into LDS
//   (STAGE, rows up to 32768 bins) and every scan of row_features_core.h runs on the LDS copy (rows of <= 4096
//   bins: on a register copy of that): 4 B/bin of HBM traffic.  Longer rows are scanned in place (L2-resident).
//   Three builds instead of one kernel that branches on the length: rows of exactly 4096 bins (the reference's
//   frame; bounds checks fold, four workgroups per SIMD: 30 -> 22.6 ns per row), other short rows, long rows.
// row_peaks_kernel            the peak scan alone, for caller-supplied thresholds (sdrk_row_peaks).
// The N = 4096 transform can also run the same routine as its epilogue (fft4096.hip, EPI_FEATURES).
#include "row_features_core.h"
#include "../../include/sdrk.h"   // SDRK_FEAT_* plane numbers

namespace sdrk {

constexpr int RF_STAGE_MAX = 32768;   // bins: 128 KiB of the 160 KiB LDS

// SHORT rows (<= 4096 bins: the register-resident routine) are built for four workgroups per SIMD (<= 128 VGPRs, as
// the fused kernel), longer rows (LDS / L2 scans, few registers needed, LDS-limited anyway) without a bound.
// NCONST: the row length as a compile-time constant (4096, the reference's: bounds checks fold, 123 VGPRs and no
// scratch) or 0 = the argument (128 VGPRs, four spilled registers).
template <bool STAGE, bool SHORT, int NCONST>
__global__ __launch_bounds__(RF_THREADS, SHORT ? 4 : 2) void row_features_kernel(const float* __restrict__ rows, size_t n_rows, int nfft,
                                                                 RowFeatParams prm, double* __restrict__ stats,
                                                                 double* __restrict__ thr, int* __restrict__ idx,
                                                                 int* __restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) float rf_row[];
    __shared__ RowFeatShared sh;
    if (NCONST) nfft = NCONST;
    for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        // (re-read per row behind an opaque copy: hoisted out of this persistent loop, the staging addresses derived
        // from the thread number cost registers that the reductions then spill)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < RF_THREADS);
        const float* __restrict__ x = rows + r * (size_t)nfft;
        double* o_thr = thr ? thr + r : nullptr;
        int* o_idx = idx ? idx + r * (size_t)prm.max_peaks : nullptr;
        int* o_cnt = cnt ? cnt + r : nullptr;
        if (STAGE) {
            if ((nfft & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
                typedef float rf_v4f __attribute__((ext_vector_type(4)));
                const rf_v4f* __restrict__ x4 = reinterpret_cast<const rf_v4f*>(x);
                rf_v4f* __restrict__ l4 = reinterpret_cast<rf_v4f*>(rf_row);
                for (int i = tid; i < nfft / 4; i += RF_THREADS) l4[i] = __builtin_nontemporal_load(&x4[i]);
            } else {
#pragma unroll 1                                         // (a caller's oddly aligned rows: unrolled, its sixteen hoisted offsets spill)
                for (int i = tid; i < nfft; i += RF_THREADS) rf_row[i] = x[i];
            }
            __syncthreads();
            row_features_wg<SHORT ? 1 : 2>(rf_row, nfft, prm, sh, stats + r * 16, o_thr, o_idx, o_cnt);
        } else {
            row_features_wg<SHORT ? 1 : 2>(x, nfft, prm, sh, stats + r * 16, o_thr, o_idx, o_cnt);
        }
        __syncthreads();   // the staged row and the shared scratch are reused by the next row
    }
}

__global__ __launch_bounds__(RF_THREADS) void row_peaks_kernel(const float* __restrict__ rows, int nfft,
                                                               const double* __restrict__ thresholds,
                                                               int min_distance, int max_peaks,
                                                               int* __restrict__ out_idx, int* __restrict__ out_count) {
    __shared__ unsigned long long flags[4];
    __shared__ int sh_state[2];  // last accepted index, count
    const float* __restrict__ x = rows + (size_t)blockIdx.x * nfft;
    const double thr = thresholds[blockIdx.x];
    int* __restrict__ idx = out_idx + (size_t)blockIdx.x * max_peaks;
    const int tid = threadIdx.x;
    if (tid == 0) { sh_state[0] = -min_distance; sh_state[1] = 0; }
    __syncthreads();
    for (int base = 0; base < nfft; base += RF_THREADS) {
        const int i = base + tid;
        bool cand = false;
        if (i >= 1 && i < nfft - 1) {
            const double v = (double)x[i];
            cand = v > thr && x[i] > x[i - 1] && x[i] > x[i + 1];
        }
        const unsigned long long b = __ballot(cand);
        if ((tid & 63) == 0) flags[tid >> 6] = b;
        __syncthreads();
        if (tid == 0) {
            int last = sh_state[0], count = sh_state[1];
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = flags[w];
                const int w0 = base + 64 * w;
                m = rf_clear_below(m, last + min_distance - w0);     // see row_features_core.h
                while (m) {
                    const int bit = __ffsll((long long)m) - 1;
                    last = w0 + bit;
                    if (count < max_peaks) idx[count] = last;
                    ++count;
                    m = rf_clear_below(m, bit + min_distance);
                }
            }
            sh_state[0] = last; sh_state[1] = count;
        }
        __syncthreads();
    }
    if (tid == 0) out_count[blockIdx.x] = sh_state[1];
}

// The per-row arithmetic the reference does AFTER its helpers' scans, for a whole batch at once (round 4): what
// features._assemble_arrays used to do in numpy over the packed results (12-14 ms per 32768 frames, 40 % of the host
// call) now happens here, one wave per row, and the host receives finished planes (sdrk.h: SDRK_FEAT_*):
//   noise floor  = numpy.percentile's float32 interpolation between sorted[rank], sorted[rank+1]  (classifier.py:179-181)
//   snr          = float32(max - noise floor)                                                      (:46)
//   flatness     = clip(exp(mean ln p) / mean p, 0, 1)                                             (:186-189)
//   kurtosis     = 0 if sigma < 1e-9 else m4 / m2^2                                                (:195-198)
//   bandwidths   = f[last] - f[first] of the bins within 3 / 10 / 20 dB of the maximum, 0 when there is none (:163-170)
//   peak spacing = std(diff(f[peaks])) over the kept peaks, 0 for fewer than three                  (:214-219)
//   peak density = peak_count / nfft
// Every float32 step is written with the rf_*_f32 helpers (no contraction), as in rf_finish.
__global__ __launch_bounds__(256) void feature_finalize_kernel(const double* __restrict__ stats, const double* __restrict__ thr,
                                                               const int* __restrict__ idx, const int* __restrict__ cnt,
                                                               size_t n_rows, int nfft, float gamma, int max_peaks,
                                                               const double* __restrict__ freqs, double* __restrict__ out) {
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const double* __restrict__ s = stats + r * 16;
    long long* __restrict__ outi = reinterpret_cast<long long*>(out);
    const int k_all = cnt ? cnt[r] : 0;
    const int k = k_all < max_peaks ? k_all : max_peaks;
    // peak spacing: two passes over the <= max_peaks kept indices, lanes striding the slots, float64 wave sums
    double spacing = 0.0;
    if (freqs && idx && k >= 3) {
        const int* __restrict__ pk = idx + r * (size_t)max_peaks;
        const int m = k - 1;
        double acc = 0.0;
        for (int j = lane; j < m; j += 64) acc += freqs[pk[j + 1]] - freqs[pk[j]];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        const double mean = acc / m;
        double v = 0.0;
        for (int j = lane; j < m; j += 64) {
            const double d = (freqs[pk[j + 1]] - freqs[pk[j]]) - mean;
            v += d * d;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        spacing = sqrt(v / m);
    }
    if (lane != 0) return;
    const float mx = (float)s[0], q0 = (float)s[1], q1 = (float)s[2];
    const float diff = rf_sub_f32(q1, q0);
    float nf = rf_add_f32(q0, rf_mul_f32(diff, gamma));
    if (gamma >= 0.5f) nf = rf_sub_f32(q1, rf_mul_f32(diff, rf_sub_f32(1.0f, gamma)));
    out[SDRK_FEAT_MAX_DB * n_rows + r] = (double)mx;
    out[SDRK_FEAT_NOISE_FLOOR_DB * n_rows + r] = (double)nf;
    out[SDRK_FEAT_SNR_DB * n_rows + r] = (double)rf_sub_f32(mx, nf);
    double fl = exp(s[6]) / s[7];
    fl = fl < 0.0 ? 0.0 : (fl > 1.0 ? 1.0 : fl);                                   // (NaN stays NaN, as np.clip leaves it)
    out[SDRK_FEAT_FLATNESS * n_rows + r] = fl;
    out[SDRK_FEAT_KURTOSIS * n_rows + r] = sqrt(s[4]) < 1e-9 ? 0.0 : s[5] / (s[4] * s[4]);
    out[SDRK_FEAT_THRESHOLD_DB * n_rows + r] = thr ? thr[r] : 0.0;
    out[SDRK_FEAT_PEAK_SPACING_STD_HZ * n_rows + r] = spacing;
    out[SDRK_FEAT_PEAK_DENSITY * n_rows + r] = (double)k_all / (double)(nfft > 1 ? nfft : 1);
    outi[SDRK_FEAT_ARGMAX * n_rows + r] = (long long)s[14];
    outi[SDRK_FEAT_PEAK_COUNT * n_rows + r] = k_all;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const long long lo = (long long)s[8 + 2 * j], hi = (long long)s[9 + 2 * j];
        outi[(SDRK_FEAT_OCCUPIED_BINS + 2 * j) * n_rows + 2 * r] = lo;
        outi[(SDRK_FEAT_OCCUPIED_BINS + 2 * j) * n_rows + 2 * r + 1] = hi;
        // an all-NaN row has no bin >= max - x: the scans return first > last, the reference 0.0 (:166-168)
        const bool ok = freqs && 0 <= lo && lo <= hi && hi < nfft;
        out[(SDRK_FEAT_BANDWIDTH_HZ + j) * n_rows + r] = ok ? freqs[hi] - freqs[lo] : 0.0;
    }
}

hipError_t launch_feature_finalize(const double* d_stats, const double* d_thr, const int* d_idx, const int* d_cnt,
                                   size_t n_rows, int nfft, float gamma, int max_peaks, const double* d_freqs,
                                   double* d_out, hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(feature_finalize_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, d_stats, d_thr, d_idx,
                       d_cnt, n_rows, nfft, gamma, max_peaks, d_freqs, d_out);
    return hipGetLastError();
}

hipError_t launch_row_features(const float* d_rows, size_t n_rows, int nfft, int rank, float gamma, int min_distance,
                               int max_peaks, double* d_stats, double* d_thr, int* d_idx, int* d_cnt, int num_cus,
                               hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    RowFeatParams prm{rank, gamma, min_distance, max_peaks};
    const bool stage = nfft <= RF_STAGE_MAX;
    const size_t lds_bytes = stage ? (size_t)((nfft + 3) & ~3) * sizeof(float) : 0;
    size_t per_cu = stage ? (150 * 1024) / (lds_bytes + sizeof(RowFeatShared)) : 8;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const size_t cap = (size_t)num_cus * per_cu;
    const unsigned grid = (unsigned)(n_rows < cap ? n_rows : cap);
    if (nfft == 4096) {
        hipLaunchKernelGGL((row_features_kernel<true, true, 4096>), dim3(grid), dim3(RF_THREADS), lds_bytes, s, d_rows, n_rows, nfft,
                           prm, d_stats, d_thr, d_idx, d_cnt);
    } else if (nfft <= 16 * RF_THREADS) {    // (always staged: <= 16 KiB)
        hipLaunchKernelGGL((row_features_kernel<true, true, 0>), dim3(grid), dim3(RF_THREADS), lds_bytes, s, d_rows, n_rows, nfft,
                           prm, d_stats, d_thr, d_idx, d_cnt);
    } else if (stage) {
        auto kern = row_features_kernel<true, false, 0>;
        static std::atomic<uint64_t> lds_ok{0};
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);
        if (e0 != hipSuccess) return e0;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RF_THREADS), lds_bytes, s, d_rows, n_rows, nfft, prm, d_stats, d_thr,
                           d_idx, d_cnt);
    } else {
        hipLaunchKernelGGL((row_features_kernel<false, false, 0>), dim3(grid), dim3(RF_THREADS), 0, s, d_rows, n_rows, nfft, prm,
                           d_stats, d_thr, d_idx, d_cnt);
    }
    return hipGetLastError();
}

hipError_t launch_row_peaks(const float* d_rows, size_t n_rows, int nfft, const double* d_thr, int min_distance,
                            int max_peaks, int* d_idx, int* d_count, hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(row_peaks_kernel, dim3((unsigned)n_rows), dim3(RF_THREADS), 0, s, d_rows, nfft, d_thr,
                       min_distance, max_peaks, d_idx, d_count);
    return hipGetLastError();
}

}  // namespace sdrk
