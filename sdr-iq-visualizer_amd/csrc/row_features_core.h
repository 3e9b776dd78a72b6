// row_features_core.h — the per-row measurements of the reference's classifier helpers
// (app/processing/classifier.py:163-212) as ONE workgroup-wide routine over a power_db row that is
// already on chip (LDS), shared by the stand-alone kernel (row_features.hip: the row is read from HBM
// once and staged) and by the fused epilogue of the N = 4096 transform (fft4096.hip: the row never
// exists in HBM unless the caller also asks for it).
//
//   stats[16] (double):
//     [0] max  [1] sorted[rank]  [2] sorted[rank+1]  [3] mean  [4] mean (x-mu)^2  [5] mean (x-mu)^4
//     [6] mean ln(p)  [7] mean p   with p = max(10^(x/10), 1e-15)                       (:183-189)
//     [8],[9] first,last index with x >= max-3   [10],[11] ... max-10   [12],[13] ... max-20   (:163-170;
//     float32 compare, thresholds formed in float32 as numpy forms them)   [14] argmax (first)   [15] n
//   thr: the adaptive peak threshold max(noise_floor + 5, max - 0.9 snr + 5) of :55 with numpy's dtype
//     rules (float32 percentile interpolation, NEP-50 scalar promotion) — see features.py, which
//     recomputes it on the host from stats[] and must agree to the bit
//   peaks: strict local maxima above thr, accepted left to right when >= min_distance bins after the
//     previously accepted one (:200-212)
//
// Order statistics: exact radix select on the float32 keys, 8 bits per pass (no sort); the second one
// (rank+1) costs one more pass: it equals sorted[rank] when more than rank+1 elements are <= it, else it
// is the smallest larger element.
#pragma once
#include "kernels.h"

namespace sdrk {

constexpr int RF_THREADS = 256;

struct RowFeatShared {
    double d[4 * 3];
    float f[4];
    int i[4 * 7];
    unsigned hist[256];
    unsigned state[4];
    unsigned long long flags[64];   // candidate bits of one 4096-bin stretch of the row
    int pk[2];
    int list[64];                   // accepted peak indices waiting for one coalesced store
    double thr;
};

struct RowFeatParams {
    int rank;           // floor(float32(n-1) * float32(q)/100): numpy's lower index of the q-th percentile
    float gamma;        // its float32 interpolation weight
    int min_distance;   // peak spacing, bins
    int max_peaks;      // capacity of the per-row index list
};

__device__ __forceinline__ unsigned rf_key(float x) {  // order-preserving map to unsigned (NaNs last, as numpy sorts)
    unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rf_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// m with bits [0, n) cleared (n may be <= 0 or >= 64)
__device__ __forceinline__ unsigned long long rf_clear_below(unsigned long long m, int n) {
    return n <= 0 ? m : (n >= 64 ? 0ull : m & (~0ull << n));
}

// 10^(v/10) in double to ~1e-11 relative (the sums it feeds are compared at 1e-9): 2^t with t = v log2(10)/10
// split into an integer and a fraction in [-1/2, 1/2], degree-10 Taylor polynomial of e^(f ln 2), v_ldexp_f64.
// libm's pow() + log() per element (the literal restatement) cost ~6x this; ln p needs no transcendental at all:
// ln(10^(v/10)) = v ln(10)/10.
__device__ __forceinline__ double rf_pow10_tenth(double v) {
    const double t = v * 0.33219280948873623479;   // log2(10) / 10
    const double k = rint(t);
    const double g = (t - k) * 0.69314718055994530942;
    double p = 2.7557319223985890653e-07;           // 1/10!
    p = fma(p, g, 2.7557319223985892511e-06);       // 1/9!
    p = fma(p, g, 2.4801587301587301566e-05);       // 1/8!
    p = fma(p, g, 1.9841269841269841253e-04);       // 1/7!
    p = fma(p, g, 1.3888888888888889419e-03);       // 1/6!
    p = fma(p, g, 8.3333333333333332177e-03);       // 1/5!
    p = fma(p, g, 4.1666666666666664354e-02);       // 1/4!
    p = fma(p, g, 1.6666666666666665741e-01);       // 1/3!
    p = fma(p, g, 0.5);
    p = fma(p, g, 1.0);
    p = fma(p, g, 1.0);
    const double kk = k < -1100.0 ? -1100.0 : (k > 1100.0 ? 1100.0 : k);
    return ldexp(p, (int)kk);
}

// k-th smallest (0-based) of x[0..n) and the (k+1)-th: q0, q1 (q1 == q0 when k == n-1).
// Exact radix select, 8 bits per pass.  Two things keep a pass short: (1) dB rows share their sign and high
// exponent bits, so in the top-byte pass nearly every key lands in the same 1-3 bins — that pass counts equal
// bins inside each wave first (ballot per distinct value) and issues one LDS atomic per value instead of 64
// serialised same-address ones (the lower bytes are spread and use plain atomics); (2) the digit that holds the wanted rank is found by a 256-thread prefix
// scan over the histogram, not by one thread walking 256 LDS words.
template <class RowPtr>
__device__ __forceinline__ void rf_select_pair(RowPtr x, int n, unsigned rank, RowFeatShared& sh, float& q0, float& q1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rows of up to 4096 bins (the reference's frame length): the 16 keys of a thread stay in registers for all
    // passes; 0 is the key of no float (-NaN with every payload bit set aside), used for the slots past n
    const bool cached = n <= 16 * RF_THREADS;
    unsigned kreg[16];
    if (cached) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = tid + RF_THREADS * j;
            kreg[j] = i < n ? rf_key(x[i]) : 0u;
        }
    }
    unsigned prefix = 0, mask = 0, want = rank;
    for (int shift = 24; shift >= 0; shift -= 8) {
        __syncthreads();
        sh.hist[tid] = 0;
        __syncthreads();
        auto count_key = [&](unsigned k, bool valid) {
            const bool in = valid && (k & mask) == prefix;
            const unsigned bin = (k >> shift) & 255u;
            if (shift == 24) {
                unsigned long long todo = __ballot(in);
                while (todo) {                                   // one iteration per distinct bin in the wave
                    const int leader = __ffsll((long long)todo) - 1;
                    const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                    const unsigned long long same = __ballot(in && bin == b);
                    if (lane == leader) atomicAdd(&sh.hist[b], (unsigned)__popcll(same));
                    todo &= ~same;
                }
            } else if (in) {
                atomicAdd(&sh.hist[bin], 1u);
            }
        };
        if (cached) {
#pragma unroll
            for (int j = 0; j < 16; ++j) count_key(kreg[j], tid + RF_THREADS * j < n);
        } else {
            for (int i0 = 0; i0 < n; i0 += RF_THREADS) {
                const int i = i0 + tid;
                count_key(i < n ? rf_key(x[i]) : 0u, i < n);
            }
        }
        __syncthreads();
        // inclusive scan of the 256 bins: thread t owns bin t
        const unsigned h = sh.hist[tid];
        unsigned incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) sh.state[wave] = incl;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wave; ++w) before += sh.state[w];
        incl += before;
        __syncthreads();
        if (want >= incl - h && want < incl) {                   // exactly one thread: its bin holds the rank
            sh.state[0] = (unsigned)tid;
            sh.state[1] = want - (incl - h);
        }
        __syncthreads();
        prefix |= sh.state[0] << shift;
        mask |= 255u << shift;
        want = sh.state[1];
    }
    // one more scan: how many keys are <= prefix, and the smallest key above it
    unsigned le = 0, next = 0xFFFFFFFFu;
    if (cached) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (tid + RF_THREADS * j < n) {
                if (kreg[j] <= prefix) ++le;
                else next = kreg[j] < next ? kreg[j] : next;
            }
        }
    } else {
        for (int i = tid; i < n; i += RF_THREADS) {
            const unsigned k = rf_key(x[i]);
            if (k <= prefix) ++le;
            else next = k < next ? k : next;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        le += __shfl_down(le, o, 64);
        const unsigned on = __shfl_down(next, o, 64);
        next = on < next ? on : next;
    }
    __syncthreads();
    if (lane == 0) { sh.hist[wave] = le; sh.hist[4 + wave] = next; }
    __syncthreads();
    le = sh.hist[0] + sh.hist[1] + sh.hist[2] + sh.hist[3];
    next = sh.hist[4];
    for (int w = 1; w < 4; ++w) next = sh.hist[4 + w] < next ? sh.hist[4 + w] : next;
    q0 = rf_unkey(prefix);
    q1 = (le > rank + 1 || next == 0xFFFFFFFFu) ? q0 : rf_unkey(next);
    __syncthreads();
}

// The whole measurement of one row by one 256-thread workgroup.  `x` points at n float32 values (LDS or global).
// o_stats: 16 doubles; o_thr: 1 double; o_idx: max_peaks ints; o_cnt: 1 int (total peaks found, may exceed max_peaks).
// o_idx / o_cnt / o_thr may be null (stats only).
template <class RowPtr>
__device__ __forceinline__ void row_features_wg(RowPtr x, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                                double* __restrict__ o_stats, double* __restrict__ o_thr,
                                                int* __restrict__ o_idx, int* __restrict__ o_cnt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // scan A: max (+ first argmax), sums for mean and flatness
    float mx = -INFINITY;
    int amx = 0x7fffffff;
    double sx = 0.0, sp = 0.0, slp = 0.0;
    for (int i = tid; i < n; i += RF_THREADS) {
        const float v = x[i];
        if (v > mx) { mx = v; amx = i; }
        sx += (double)v;
        double p = v == -INFINITY ? 0.0 : rf_pow10_tenth((double)v);   // a row computed with eps = 0 can hold -inf
        double lp = (double)v * 0.23025850929940456840;      // ln(10) / 10
        if (p < 1e-15) { p = 1e-15; lp = -34.538776394910684; }       // np.clip(p, 1e-15, None), then log (NaN stays NaN)
        sp += p;
        slp += lp;
    }
    // one shuffle tree for all five partials (independent ds_bpermutes overlap), one LDS hand-off between the waves
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_down(mx, off, 64);
        const int oi = __shfl_down(amx, off, 64);
        sx += __shfl_down(sx, off, 64);
        sp += __shfl_down(sp, off, 64);
        slp += __shfl_down(slp, off, 64);
        if (om > mx || (om == mx && oi < amx)) { mx = om; amx = oi; }
    }
    __syncthreads();
    if (lane == 0) { sh.f[wave] = mx; sh.i[wave] = amx; sh.d[wave * 3] = sx; sh.d[wave * 3 + 1] = sp; sh.d[wave * 3 + 2] = slp; }
    __syncthreads();
    mx = sh.f[0]; amx = sh.i[0];
    for (int w = 1; w < 4; ++w)
        if (sh.f[w] > mx || (sh.f[w] == mx && sh.i[w] < amx)) { mx = sh.f[w]; amx = sh.i[w]; }
    const double mean = (sh.d[0] + sh.d[3] + sh.d[6] + sh.d[9]) / n;
    const double mean_p = (sh.d[1] + sh.d[4] + sh.d[7] + sh.d[10]) / n;
    const double mean_lp = (sh.d[2] + sh.d[5] + sh.d[8] + sh.d[11]) / n;

    // scan B: central moments, occupied-band edges (thresholds in float32, as peak - float(drop) is)
    const float t3 = mx - 3.0f, t10 = mx - 10.0f, t20 = mx - 20.0f;
    double s2 = 0.0, s4 = 0.0;
    int f3 = 0x7fffffff, l3 = -1, f10 = 0x7fffffff, l10 = -1, f20 = 0x7fffffff, l20 = -1;
    for (int i = tid; i < n; i += RF_THREADS) {
        const float v = x[i];
        const double dv = (double)v - mean, d2 = dv * dv;
        s2 += d2;
        s4 += d2 * d2;
        if (v >= t3) { f3 = min(f3, i); l3 = max(l3, i); }
        if (v >= t10) { f10 = min(f10, i); l10 = max(l10, i); }
        if (v >= t20) { f20 = min(f20, i); l20 = max(l20, i); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s2 += __shfl_down(s2, off, 64);
        s4 += __shfl_down(s4, off, 64);
        f3 = min(f3, __shfl_down(f3, off, 64));   l3 = max(l3, __shfl_down(l3, off, 64));
        f10 = min(f10, __shfl_down(f10, off, 64)); l10 = max(l10, __shfl_down(l10, off, 64));
        f20 = min(f20, __shfl_down(f20, off, 64)); l20 = max(l20, __shfl_down(l20, off, 64));
    }
    __syncthreads();
    if (lane == 0) {
        int* s = sh.i + 4 + wave * 6;
        s[0] = f3; s[1] = l3; s[2] = f10; s[3] = l10; s[4] = f20; s[5] = l20;
        sh.d[wave * 3] = s2; sh.d[wave * 3 + 1] = s4;
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        const int* s = sh.i + 4 + w * 6;
        f3 = min(f3, s[0]); l3 = max(l3, s[1]); f10 = min(f10, s[2]); l10 = max(l10, s[3]);
        f20 = min(f20, s[4]); l20 = max(l20, s[5]);
    }
    const double m2 = (sh.d[0] + sh.d[3] + sh.d[6] + sh.d[9]) / n;
    const double m4 = (sh.d[1] + sh.d[4] + sh.d[7] + sh.d[10]) / n;

    // order statistics for numpy.percentile's linear interpolation
    const int r0 = prm.rank < 0 ? 0 : (prm.rank > n - 1 ? n - 1 : prm.rank);
    float q0, q1;
    rf_select_pair(x, n, (unsigned)r0, sh, q0, q1);

    if (tid == 0) {
        double* o = o_stats;
        o[0] = mx; o[1] = q0; o[2] = q1; o[3] = mean; o[4] = m2; o[5] = m4; o[6] = mean_lp; o[7] = mean_p;
        o[8] = f3; o[9] = l3; o[10] = f10; o[11] = l10; o[12] = f20; o[13] = l20; o[14] = amx; o[15] = n;
        // numpy.percentile on a float32 row: a + (b-a)*gamma, or b - (b-a)*(1-gamma) when gamma >= 0.5, every
        // operation rounded to float32 (no contraction); then classifier.py:46,55 with NEP-50 promotion:
        // snr = float32(max - nf) widened; second = (max - float32(0.9*snr)) + 5 in float32; first = nf + 5 in
        // float64; python max(first, second) compares second > float32(first).
        const float diff = __fsub_rn(q1, q0);
        float nf = __fadd_rn(q0, __fmul_rn(diff, prm.gamma));
        if (prm.gamma >= 0.5f) nf = __fsub_rn(q1, __fmul_rn(diff, __fsub_rn(1.0f, prm.gamma)));
        const double snr = (double)__fsub_rn(mx, nf);
        const float second = __fadd_rn(__fsub_rn(mx, (float)(0.9 * snr)), 5.0f);
        const double first = (double)nf + 5.0;
        sh.thr = second > (float)first ? (double)second : first;
        if (o_thr) *o_thr = sh.thr;
        sh.pk[0] = -prm.min_distance;
        sh.pk[1] = 0;
    }
    __syncthreads();
    if (!o_idx || !o_cnt) return;

    // greedy peak scan.  All four waves mark the candidates of a 4096-bin stretch (64 ballot words in LDS), then wave 0
    // applies the spacing rule in index order: two barriers per stretch instead of two per 256 bins.
    const double thr = sh.thr;
    for (int sbase = 0; sbase < n; sbase += 4096) {
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            const int i = sbase + 256 * c + tid;
            bool cand = false;
            if (i >= 1 && i < n - 1) {
                const float v = x[i];
                cand = (double)v > thr && v > x[i - 1] && v > x[i + 1];
            }
            const unsigned long long b = __ballot(cand);
            if (lane == 0) sh.flags[4 * c + wave] = b;
        }
        __syncthreads();
        if (wave == 0) {
            // every lane of wave 0 runs the same scalar recurrence (SGPR operands: the flag words and the running
            // state are wave-uniform).  The 64 flag words are fetched by one LDS read (lane w holds word w) and
            // handed out by v_readlane; accepted indices collect in an LDS list that leaves as one store per 64.
            int last = __builtin_amdgcn_readfirstlane(sh.pk[0]), count = __builtin_amdgcn_readfirstlane(sh.pk[1]);
            int fill = 0;                                            // entries waiting in sh.list; count = stored so far
            const unsigned long long dspan = prm.min_distance >= 64 ? ~0ull : (1ull << (prm.min_distance > 0 ? prm.min_distance : 1)) - 1ull;
            const int words = (n - sbase + 63) / 64 < 64 ? (n - sbase + 63) / 64 : 64;
            const unsigned long long mine = sh.flags[lane];
            const int mine_lo = (int)(unsigned)mine, mine_hi = (int)(unsigned)(mine >> 32);
            auto flush = [&]() {
                if (lane < fill && count + lane < prm.max_peaks) o_idx[count + lane] = sh.list[lane];
                count += fill;
                fill = 0;
            };
            for (int w = 0; w < words; ++w) {
                unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mine_hi, w) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane(mine_lo, w);
                const int w0 = sbase + 64 * w;
                // candidates closer than min_distance to the last accepted peak are rejected wholesale: one short
                // scalar iteration per ACCEPTED peak (noise rows have ~3 candidates per accepted one) that only
                // marks it; the marked bits are turned into list entries by all 64 lanes at once afterwards
                m = rf_clear_below(m, last + prm.min_distance - w0);
                unsigned long long acc = 0;
                while (m) {                  // dependent chain per accepted peak: find-first-one, shift, and-not
                    const int bit = __builtin_ctzll(m);
                    acc |= 1ull << bit;
                    m &= ~(dspan << bit);
                }
                if (acc) {
                    last = w0 + 63 - __builtin_clzll(acc);
                    const int pc = __popcll(acc);
                    if (fill + pc > 64) flush();
                    if ((acc >> lane) & 1ull) sh.list[fill + __popcll(acc & ((1ull << lane) - 1ull))] = w0 + lane;
                    fill += pc;
                }
            }
            flush();
            if (lane == 0) { sh.pk[0] = last; sh.pk[1] = count; }
        }
        __syncthreads();
    }
    if (tid == 0) *o_cnt = sh.pk[1];
}

}  // namespace sdrk
