// row_features.hip — per-row reductions over a power_db row, on the device, so that
// the immediate consumer of every spectrum row does not force a D2H of the row.
//
// These are the O(N) measurements behind the reference's rule-based classifier
// (app/processing/classifier.py): max and 20th-percentile noise floor (:45-46,179-181),
// occupied bandwidth at -3/-10/-20 dB (:163-170), spectral flatness (:183-189), kurtosis
// (:191-198) and the greedy peak list (:200-212).  The rule ladder (:69-122) and the
// 12-frame smoothing (:125-139) are scalar application logic and stay with the caller.
//
// row_stats_kernel   one workgroup per row -> 16 doubles:
//   [0] max  [1] sorted[rank]  [2] sorted[rank+1]  [3] mean  [4] mean (x-mu)^2  [5] mean (x-mu)^4
//   [6] mean ln(p)  [7] mean p   with p = max(10^(x/10), 1e-15)
//   [8],[9] first,last index with x >= max-3   [10],[11] … max-10   [12],[13] … max-20   (float32 compare,
//   thresholds formed in float32 as numpy forms them)   [14] argmax (first)   [15] n
//   Order statistics by an exact 4-pass radix select on the float32 keys (no sort).
// row_peaks_kernel   one workgroup per row: strict local maxima above a float64 threshold, accepted
//   left to right when at least min_distance bins after the previously accepted one.
#include "kernels.h"

namespace sdrk {

constexpr int RF_THREADS = 256;

__device__ __forceinline__ double wg_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ unsigned f32_key(float x) {  // order-preserving map to unsigned
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// k-th smallest (0-based) of x[0..n): exact radix select, 8 bits per pass.
__device__ float wg_select(const float* __restrict__ x, int n, unsigned rank, unsigned* hist, unsigned* state) {
    unsigned prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        __syncthreads();
        hist[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += RF_THREADS) {
            const unsigned k = f32_key(x[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned acc = 0, b = 0;
            for (; b < 256; ++b) {
                if (acc + hist[b] > rank) break;
                acc += hist[b];
            }
            state[0] = b;
            state[1] = rank - acc;
        }
        __syncthreads();
        prefix |= state[0] << shift;
        mask |= 255u << shift;
        rank = state[1];
    }
    return key_f32(prefix);
}

__global__ __launch_bounds__(RF_THREADS) void row_stats_kernel(const float* __restrict__ rows, int nfft, int rank,
                                                               double* __restrict__ out) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    __shared__ int shi[4 * 7];
    __shared__ unsigned hist[256];
    __shared__ unsigned state[2];
    const float* __restrict__ x = rows + (size_t)blockIdx.x * nfft;
    double* __restrict__ o = out + (size_t)blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // scan A: max (+ first argmax), sums for mean and flatness
    float mx = -INFINITY;
    int amx = 0x7fffffff;
    double sx = 0.0, sp = 0.0, slp = 0.0;
    for (int i = tid; i < nfft; i += RF_THREADS) {
        const float v = x[i];
        if (v > mx) { mx = v; amx = i; }
        sx += (double)v;
        double p = pow(10.0, (double)v / 10.0);
        p = p < 1e-15 ? 1e-15 : p;
        sp += p;
        slp += log(p);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_down(mx, off, 64);
        const int oi = __shfl_down(amx, off, 64);
        if (om > mx || (om == mx && oi < amx)) { mx = om; amx = oi; }
    }
    if (lane == 0) { shf[wave] = mx; shi[wave] = amx; }
    __syncthreads();
    mx = shf[0]; amx = shi[0];
    for (int w = 1; w < 4; ++w)
        if (shf[w] > mx || (shf[w] == mx && shi[w] < amx)) { mx = shf[w]; amx = shi[w]; }
    const double mean = wg_sum(sx, shd) / nfft;
    const double mean_p = wg_sum(sp, shd) / nfft;
    const double mean_lp = wg_sum(slp, shd) / nfft;

    // scan B: central moments, occupied-band edges (thresholds in float32, as peak - float(drop) is)
    const float t3 = mx - 3.0f, t10 = mx - 10.0f, t20 = mx - 20.0f;
    double s2 = 0.0, s4 = 0.0;
    int f3 = 0x7fffffff, l3 = -1, f10 = 0x7fffffff, l10 = -1, f20 = 0x7fffffff, l20 = -1;
    for (int i = tid; i < nfft; i += RF_THREADS) {
        const float v = x[i];
        const double d = (double)v - mean, d2 = d * d;
        s2 += d2;
        s4 += d2 * d2;
        if (v >= t3) { f3 = min(f3, i); l3 = max(l3, i); }
        if (v >= t10) { f10 = min(f10, i); l10 = max(l10, i); }
        if (v >= t20) { f20 = min(f20, i); l20 = max(l20, i); }
    }
    const double m2 = wg_sum(s2, shd) / nfft;
    const double m4 = wg_sum(s4, shd) / nfft;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        f3 = min(f3, __shfl_down(f3, off, 64));   l3 = max(l3, __shfl_down(l3, off, 64));
        f10 = min(f10, __shfl_down(f10, off, 64)); l10 = max(l10, __shfl_down(l10, off, 64));
        f20 = min(f20, __shfl_down(f20, off, 64)); l20 = max(l20, __shfl_down(l20, off, 64));
    }
    __syncthreads();
    if (lane == 0) {
        int* s = shi + 4 + wave * 6;
        s[0] = f3; s[1] = l3; s[2] = f10; s[3] = l10; s[4] = f20; s[5] = l20;
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        const int* s = shi + 4 + w * 6;
        f3 = min(f3, s[0]); l3 = max(l3, s[1]); f10 = min(f10, s[2]); l10 = max(l10, s[3]);
        f20 = min(f20, s[4]); l20 = max(l20, s[5]);
    }

    // order statistics
    const int r0 = rank < 0 ? 0 : (rank > nfft - 1 ? nfft - 1 : rank);
    const int r1 = r0 + 1 > nfft - 1 ? nfft - 1 : r0 + 1;
    const float q0 = wg_select(x, nfft, (unsigned)r0, hist, state);
    const float q1 = wg_select(x, nfft, (unsigned)r1, hist, state);

    if (tid == 0) {
        o[0] = mx; o[1] = q0; o[2] = q1; o[3] = mean; o[4] = m2; o[5] = m4; o[6] = mean_lp; o[7] = mean_p;
        o[8] = f3; o[9] = l3; o[10] = f10; o[11] = l10; o[12] = f20; o[13] = l20; o[14] = amx; o[15] = nfft;
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
                while (m) {
                    const int bit = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int p = base + 64 * w + bit;
                    if (p - last >= min_distance) {
                        if (count < max_peaks) idx[count] = p;
                        ++count;
                        last = p;
                    }
                }
            }
            sh_state[0] = last; sh_state[1] = count;
        }
        __syncthreads();
    }
    if (tid == 0) out_count[blockIdx.x] = sh_state[1];
}

hipError_t launch_row_stats(const float* d_rows, size_t n_rows, int nfft, int rank, double* d_out, hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)n_rows), dim3(RF_THREADS), 0, s, d_rows, nfft, rank, d_out);
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
