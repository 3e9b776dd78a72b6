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
// Order statistics, rows of up to 4096 bins (the reference's frame length): the 16 values of a thread stay in
// registers for every scan, and the two order statistics come from ONE histogram pass — 2048 bins of 1/16 dB
// around the row's mean (a monotone float32 map, so bin order is value order), a 256-thread prefix scan to the
// bins that hold ranks r and r+1, and an exact ranking of the handful of values inside them (<= 64, else the
// radix select below takes over, as it does for rows with NaN / inf values or a percentile more than 64 dB off
// the mean).  Longer rows: the same value histogram refined level by level (rf_select_hist), with the exact radix
// select on the float32 keys, 8 bits per pass (no sort), behind it for the same exceptional rows; the second
// statistic (rank+1) costs one more pass: it equals sorted[rank] when more than rank+1 elements are <= it, else
// it is the smallest larger element.
//
// Sums: mean, variance and fourth moment in float64 exactly as before.  mean p: 10^(x/10) per bin as float32
// v_exp_f32 on an exactly reduced argument (relative error ~1e-7 per term, random sign), accumulated in float64;
// mean ln p = (ln10/10) * mean x needs no transcendental.  (BASELINE.json asks for 1e-5; the float64 degree-10
// polynomial this replaces held 1e-11 at ~6x the cost.)  A wave that holds values past 300 dB (float32 overflows
// 10^(x/10) at 385, the reference's float64 does not) sums p relative to its maximum and scales back in float64.
// The adaptive threshold's float32 steps are written under `#pragma clang fp contract(off)` (rf_mul_f32 ...):
// numpy rounds after every operation, and HIP's __fmul_rn / __fadd_rn do not stop hipcc from fusing them.
#pragma once
#include "kernels.h"

namespace sdrk {

constexpr int RF_THREADS = 256;
constexpr int RF_BINS = 2048;          // histogram select: bins of 1/RF_BINS_PER_DB dB, window [mean - 64, mean + 64) dB
constexpr float RF_BINS_PER_DB = 16.0f;
constexpr int RF_CAND = 64;            // values ranked exactly inside the target bins

struct RowFeatShared {
    double d[4 * 3];
    double d2[4 * 2];
    float f[4];
    int i[4 * 8];                   // [0,4) argmax per wave, [4,28) band edges per wave, [28,32) clipped-bin counts
    unsigned hist[256];
    unsigned state[4];
    unsigned long long flags[64];   // candidate bits of one 4096-bin stretch of the row
    int pk[2];
    int list[64];                   // accepted peak indices waiting for one coalesced store
    double thr;
    // histogram select of rows held in registers (n <= 16 * RF_THREADS)
    alignas(16) unsigned bins[RF_BINS];
    unsigned wtot[4];
    int sel[4];                     // bin of rank r, bin of rank r+1, values below the first bin, candidates collected
    float cand[RF_CAND];
    unsigned long long tbl[64];     // peak scan of short rows: exit state of every 64-bin word for every entry state
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

// 10^(v/10) in float32: t = v log2(10)/10 = k + f with k = rint(t) and f formed by two fmas (the second one
// carries the low part of the constant, so f is exact to ~1e-9), 2^f by v_exp_f32 (1 ulp), scaled by v_ldexp_f32.
// v = -inf gives 0; NaN stays NaN.
__device__ __forceinline__ float rf_pow10_tenth_f32(float v) {
    const float C_HI = 0.33219280948873623479f;                    // float32(log2(10) / 10)
    const float C_LO = (float)(0.33219280948873623479 - (double)C_HI);
    const float k = rintf(v * C_HI);
    float f = fmaf(v, C_HI, -k);
    f = fmaf(v, C_LO, f);
    const float kk = fminf(fmaxf(k, -300.0f), 300.0f);
    const float p = ldexpf(__builtin_amdgcn_exp2f(f), (int)kk);
    return v == -INFINITY ? 0.0f : p;
}

// the same for v > -149 dB (the clip-free scan of rf_small): no lower exponent clamp, no -inf case
__device__ __forceinline__ float rf_pow10_tenth_f32_inrange(float v) {
    const float C_HI = 0.33219280948873623479f;
    const float C_LO = (float)(0.33219280948873623479 - (double)C_HI);
    const float k = rintf(v * C_HI);
    float f = fmaf(v, C_HI, -k);
    f = fmaf(v, C_LO, f);
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)fminf(k, 300.0f));
}

// 10^(level/10) in float64 to ~1e-7 relative for any finite level (the scale of a sum taken relative to `level`):
// the exponent is split in float64, 2^fraction comes from v_exp_f32, v_ldexp_f64 applies the integer part
__device__ __forceinline__ double rf_pow10_tenth_f64(float level) {
    const double t = (double)level * 0.33219280948873623479;
    const double k = floor(t);
    return ldexp((double)__builtin_amdgcn_exp2f((float)(t - k)), (int)k);
}

// sum of the per-wave totals of the waves before `wave` (four waves; branch-free: written as a loop over w < wave
// the compiler does not know the trip count is <= 3 and emits an unrolled-by-8 loop with spills around it)
__device__ __forceinline__ unsigned rf_waves_before(const unsigned* t, int wave) {
    return (wave > 0 ? t[0] : 0u) + (wave > 1 ? t[1] : 0u) + (wave > 2 ? t[2] : 0u);
}

// k-th smallest (0-based) of x[0..n) and the (k+1)-th: q0, q1 (q1 == q0 when k == n-1).
// Exact radix select, 8 bits per pass.  Two things keep a pass short: (1) dB rows share their sign and high
// exponent bits, so in the top-byte pass nearly every key lands in the same 1-3 bins — that pass counts equal
// bins inside each wave first (ballot per distinct value) and issues one LDS atomic per value instead of 64
// serialised same-address ones (the lower bytes are spread and use plain atomics); (2) the digit that holds the wanted rank is found by a 256-thread prefix
// scan over the histogram, not by one thread walking 256 LDS words.
template <class RowPtr>
__device__ __forceinline__ void rf_select_pair(RowPtr x, int n, unsigned rank, RowFeatShared& sh, float& q0, float& q1, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
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
        for (int i0 = 0; i0 < n; i0 += RF_THREADS) {
            const int i = i0 + tid;
            count_key(i < n ? rf_key(x[i]) : 0u, i < n);
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
        before = rf_waves_before(sh.state, wave);
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
    for (int i = tid; i < n; i += RF_THREADS) {
        const unsigned k = rf_key(x[i]);
        if (k <= prefix) ++le;
        else next = k < next ? k : next;
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

// Wave-wide inclusive scans / reductions by DPP (gfx9 row shifts and row broadcasts: no LDS crossbar, no lane
// address registers).  After the six steps lane i holds the combination of lanes 0..i; lane 63 the wave's total.
//   row_shr:1,2,4,8 (0x111..0x118) inside each row of 16, row_bcast:15 (0x142) into rows 1 and 3, row_bcast:31
//   (0x143) into rows 2 and 3.  Lanes without a source keep `identity`.
#define RF_DPP_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
__device__ __forceinline__ unsigned rf_wave_scan_add(unsigned v) {
#define RF_STEP(CTRL, ROWS) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false);
    RF_DPP_STEPS(RF_STEP)
#undef RF_STEP
    return v;
}
__device__ __forceinline__ double rf_wave_scan_add(double v) {
#define RF_STEP(CTRL, ROWS)                                                                                   \
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false),          \
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false));
    RF_DPP_STEPS(RF_STEP)
#undef RF_STEP
    return v;
}
// (value, index) with "larger value, then smaller index" wins: lane 63 ends up with the wave's max and its first index
__device__ __forceinline__ float rf_wave_scan_max(float mx) {
#define RF_STEP(CTRL, ROWS) \
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xff800000, __builtin_bit_cast(int, mx), CTRL, ROWS, 0xf, false)));
    RF_DPP_STEPS(RF_STEP)
#undef RF_STEP
    return mx;
}
__device__ __forceinline__ double rf_lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Redefine sixteen registers as far as the compiler can tell (no instruction): values derived from them earlier
// are recomputed where they are needed again instead of being kept alive across a phase.
__device__ __forceinline__ void rf_opaque(float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(v[j]));
}

// Everything scans A / B and the select produced, for the common tail.
// k-th smallest of a row that is read where it lies (LDS or L2 / HBM), and the next one up: q0, q1.
// A histogram of the VALUE, as rf_small's, refined level by level: level 0 has 2048 bins of 1/16 dB around the
// row's mean; while the bin that holds rank k has more than RF_CAND members, the next level splits THAT bin into
// 2048 (t' = (t - b) * 2048: exact in float32, monotone), up to three levels (1.5e-8 dB: below float32 spacing).
// The members of the last bin are then ranked exactly, as in rf_small.  One pass over the row per level, one to
// collect, one for q1 (the smallest value above q0, or q0 again when it is tied with rank k+1).  Unlike the byte
// histogram of rf_select_pair the lanes of a wave land in different bins (dB rows spread over many 1/16 dB bins but
// share their high key bytes), so the LDS atomics do not serialise.  Returns false (nothing decided) when the row
// has non-finite values (the caller passes that in), when rank k lies outside the level-0 window, or when more
// than RF_CAND values are tied: the caller then runs rf_select_pair.
template <class RowPtr>
__device__ __forceinline__ bool rf_select_hist(RowPtr x, int n, unsigned rank, float center, bool finite,
                                               RowFeatShared& sh, float& q0, float& q1, int tid) {
    if (!finite) return false;
    const int lane = tid & 63, wave = tid >> 6;
    const float off0 = -RF_BINS_PER_DB * (center - (float)(RF_BINS / 2) / RF_BINS_PER_DB);
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
    unsigned want = rank;
    int level = 0;
    // coordinate of v at level `upto` and whether v lies inside the bins chosen at the levels above it
    auto place = [&](float v, int upto, float& t) -> bool {
        t = fminf(fmaxf(fmaf(v, RF_BINS_PER_DB, off0), 0.0f), (float)(RF_BINS - 1));
        bool in = true;
        if (upto >= 1) { in = t >= p0 && t < p0 + 1.0f; t = (t - p0) * (float)RF_BINS; }
        if (upto >= 2) { in = in && t >= p1 && t < p1 + 1.0f; t = (t - p1) * (float)RF_BINS; }
        if (upto >= 3) { in = in && t >= p2 && t < p2 + 1.0f; }
        return in;
    };
    for (;; ++level) {
        typedef unsigned rf_v4u __attribute__((ext_vector_type(4)));
        rf_v4u* b4 = reinterpret_cast<rf_v4u*>(sh.bins);
        __syncthreads();
        b4[2 * tid] = rf_v4u{0u, 0u, 0u, 0u};
        b4[2 * tid + 1] = rf_v4u{0u, 0u, 0u, 0u};
        if (tid == 0) sh.sel[3] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += RF_THREADS) {
            float t;
            if (place(x[i], level, t)) atomicAdd(&sh.bins[(int)t], 1u);
        }
        __syncthreads();
        unsigned c[8];
        {
            const rf_v4u a = b4[2 * tid], b = b4[2 * tid + 1];
            c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
        }
        const unsigned tot = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
        unsigned incl = rf_wave_scan_add(tot);
        if (lane == 63) sh.wtot[wave] = incl;
        __syncthreads();
        incl += rf_waves_before(sh.wtot, wave);
        const unsigned excl = incl - tot;
        if (excl <= want && want < incl) {            // exactly one thread: its bins hold the rank
            unsigned cum = excl;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (want >= cum && want < cum + c[k]) { sh.sel[0] = 8 * tid + k; sh.sel[1] = (int)c[k]; sh.sel[2] = (int)cum; }
                cum += c[k];
            }
        }
        __syncthreads();
        const int b = sh.sel[0], members = sh.sel[1];
        if (level == 0 && (b < 1 || b > RF_BINS - 2)) return false;     // clamped: the edge bins are open-ended
        want -= (unsigned)sh.sel[2];
        if (level == 0) p0 = (float)b; else if (level == 1) p1 = (float)b; else p2 = (float)b;
        if (members <= RF_CAND) break;
        if (level == 2) return false;                                    // > RF_CAND values tied to float32 spacing
    }
    // the members of the last bin, ranked exactly (ties by list position)
    for (int i = tid; i < n; i += RF_THREADS) {
        const float v = x[i];
        float t;
        if (place(v, level + 1, t)) {                                     // inside bins p0 .. p<level>
            const int pos = atomicAdd(&sh.sel[3], 1);
            if (pos < RF_CAND) sh.cand[pos] = v;
        }
    }
    __syncthreads();
    const int K = sh.sel[3];
    if (wave == 0) {
        const float mine = lane < K ? sh.cand[lane] : INFINITY;
        unsigned rk = 0;
        for (int j = 0; j < K; ++j) {
            const float cj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), j));
            rk += (cj < mine || (cj == mine && j < lane)) ? 1u : 0u;
        }
        if (lane < K && rk == want) sh.f[0] = mine;
    }
    __syncthreads();
    q0 = sh.f[0];
    // q1: values <= q0 number more than rank + 1 -> the next order statistic is q0 again; else the smallest above
    unsigned le = 0;
    float next = INFINITY;
    for (int i = tid; i < n; i += RF_THREADS) {
        const float v = x[i];
        if (v <= q0) ++le;
        else next = fminf(next, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        le += __shfl_down(le, o, 64);
        next = fminf(next, __shfl_down(next, o, 64));
    }
    __syncthreads();
    if (lane == 0) { sh.wtot[wave] = le; sh.cand[wave] = next; }
    __syncthreads();
    le = sh.wtot[0] + sh.wtot[1] + sh.wtot[2] + sh.wtot[3];
    next = fminf(fminf(sh.cand[0], sh.cand[1]), fminf(sh.cand[2], sh.cand[3]));
    q1 = (le > rank + 1 || rank + 1 >= (unsigned)n) ? q0 : next;
    __syncthreads();
    return true;
}

// One IEEE operation each, never part of an fma: numpy rounds after every operation, and HIP's __fmul_rn / __fadd_rn
// are plain `x * y` / `x + y` that the compiler contracts after inlining (found by tools/stress_features.py: a
// threshold one float32 ulp off in ~1 row of 60 once the surrounding code changed).  The pragma takes the `contract`
// flag off the operation where it is written, so it survives inlining.
__device__ __forceinline__ float rf_mul_f32(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float rf_add_f32(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float rf_sub_f32(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double rf_mul_f64(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}

struct RowFeatValues {
    float mx, q0, q1;
    int amx;
    double mean, m2, m4, mean_lp, mean_p;
    int f3, l3, f10, l10, f20, l20;
};

// tail shared by both row sizes: thread 0 writes stats[] and forms the adaptive threshold; returns after a barrier
__device__ __forceinline__ void rf_finish(const RowFeatValues& r, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                          double* __restrict__ o_stats, double* __restrict__ o_thr, int tid) {
    if (tid == 0) {
        double* o = o_stats;
        o[0] = r.mx; o[1] = r.q0; o[2] = r.q1; o[3] = r.mean; o[4] = r.m2; o[5] = r.m4; o[6] = r.mean_lp; o[7] = r.mean_p;
        o[8] = r.f3; o[9] = r.l3; o[10] = r.f10; o[11] = r.l10; o[12] = r.f20; o[13] = r.l20; o[14] = r.amx; o[15] = n;
        // A row that holds a NaN: np.max and np.percentile return NaN, `x >= NaN` selects no bin (0 Hz) and no peak
        // passes a NaN threshold.  The scans above skip NaNs (maximum, order statistics of the finite values), so
        // the reference's answer is put in place here.  The mean is NaN exactly when a NaN — or both infinities —
        // went into the sum; a row whose maximum is +inf keeps its own results.
        if (r.mean != r.mean && r.mx != INFINITY) {
            const double nan = r.mean;                        // (it IS NaN here; a NaN constant gets hoisted and spilled)
            o[0] = nan; o[1] = nan; o[2] = nan;
            o[8] = 0x7fffffff; o[9] = -1; o[10] = 0x7fffffff; o[11] = -1; o[12] = 0x7fffffff; o[13] = -1;
            sh.thr = nan;
        } else {
            // numpy.percentile on a float32 row: a + (b-a)*gamma, or b - (b-a)*(1-gamma) when gamma >= 0.5, every
            // operation rounded to float32 (no contraction); then classifier.py:46,55 with NEP-50 promotion:
            // snr = float32(max - nf) widened; second = (max - float32(0.9*snr)) + 5 in float32; first = nf + 5 in
            // float64; python max(first, second) compares second > float32(first).
            const float diff = rf_sub_f32(r.q1, r.q0);
            float nf = rf_add_f32(r.q0, rf_mul_f32(diff, prm.gamma));
            if (prm.gamma >= 0.5f) nf = rf_sub_f32(r.q1, rf_mul_f32(diff, rf_sub_f32(1.0f, prm.gamma)));
            const double snr = (double)rf_sub_f32(r.mx, nf);
            const float second = rf_add_f32(rf_sub_f32(r.mx, (float)rf_mul_f64(0.9, snr)), 5.0f);
            const double first = (double)nf + 5.0;
            sh.thr = second > (float)first ? (double)second : first;
        }
        if (o_thr) *o_thr = sh.thr;
        sh.pk[0] = -prm.min_distance;
        sh.pk[1] = 0;
    }
    __syncthreads();
}

template <class RowPtr>
__device__ __forceinline__ void rf_peaks(RowPtr x, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                         int* __restrict__ o_idx, int* __restrict__ o_cnt, int tid);
template <class RowPtr>
__device__ __forceinline__ void rf_peaks_small(const float (&xv)[16], RowPtr x, int n, const RowFeatParams& prm,
                                               RowFeatShared& sh, int* __restrict__ o_idx, int* __restrict__ o_cnt, int tid);

// Rows longer than 16 values per thread: every scan reads the row where it is (LDS when staged, else L2 / HBM).
template <class RowPtr>
__device__ __forceinline__ void rf_large(RowPtr x, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                         double* __restrict__ o_stats, double* __restrict__ o_thr,
                                         int* __restrict__ o_idx, int* __restrict__ o_cnt, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    RowFeatValues r;

    // scan A: max (+ first argmax), sums for mean and flatness
    float mx = -INFINITY;
    int amx = 0x7fffffff;
    double sx = 0.0, sp = 0.0, slp = 0.0;
    for (int i = tid; i < n; i += RF_THREADS) {
        const float v = x[i];
        if (v > mx) { mx = v; amx = i; }
        sx += (double)v;
        double p = (double)rf_pow10_tenth_f32(v);                       // a row computed with eps = 0 can hold -inf -> 0
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
    r.mean_p = (sh.d[1] + sh.d[4] + sh.d[7] + sh.d[10]) / n;
    r.mean_lp = (sh.d[2] + sh.d[5] + sh.d[8] + sh.d[11]) / n;
    if (mx > 300.0f) {
        // float32 overflows 10^(x/10) at 385 dB where the reference's float64 does not: sum p relative to the row's
        // maximum and scale back in float64 (one more pass over a row nobody will ever measure)
        double sq = 0.0;
        int ncl = 0;
        for (int i = tid; i < n; i += RF_THREADS) {
            const float v = x[i];
            const bool clip = v < -150.0f;
            sq += clip ? 0.0 : (double)rf_pow10_tenth_f32(v - mx);
            ncl += clip ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sq += __shfl_down(sq, off, 64);
            ncl += __shfl_down(ncl, off, 64);
        }
        __syncthreads();
        if (lane == 0) { sh.d2[wave] = sq; sh.i[28 + wave] = ncl; }
        __syncthreads();
        r.mean_p = ((sh.d2[0] + sh.d2[1] + sh.d2[2] + sh.d2[3]) * rf_pow10_tenth_f64(mx) +
                    (double)(sh.i[28] + sh.i[29] + sh.i[30] + sh.i[31]) * 1e-15) / n;
        __syncthreads();
    }

    // scan B: central moments, occupied-band edges (thresholds in float32, as peak - float(drop) is)
    const float t3 = mx - 3.0f, t10 = mx - 10.0f, t20 = mx - 20.0f;
    double s2 = 0.0, s4 = 0.0;
    int f3 = 0x7fffffff, l3 = -1, f10 = 0x7fffffff, l10 = -1, f20 = 0x7fffffff, l20 = -1;
    for (int i = tid; i < n; i += RF_THREADS) {
        const float v = x[i];
        const double dv = (double)v - mean, d2 = rf_mul_f64(dv, dv);   // (as in rf_small: one product, one sum, one written-out fma)
        s2 += d2;
        s4 = fma(d2, d2, s4);
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
    if (lane == 0) {
        int* s = sh.i + 4 + wave * 6;
        s[0] = f3; s[1] = l3; s[2] = f10; s[3] = l10; s[4] = f20; s[5] = l20;
        sh.d2[wave * 2] = s2; sh.d2[wave * 2 + 1] = s4;
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        const int* s = sh.i + 4 + w * 6;
        f3 = min(f3, s[0]); l3 = max(l3, s[1]); f10 = min(f10, s[2]); l10 = max(l10, s[3]);
        f20 = min(f20, s[4]); l20 = max(l20, s[5]);
    }
    r.m2 = (sh.d2[0] + sh.d2[2] + sh.d2[4] + sh.d2[6]) / n;
    r.m4 = (sh.d2[1] + sh.d2[3] + sh.d2[5] + sh.d2[7]) / n;
    r.mx = mx; r.amx = amx; r.mean = mean;
    r.f3 = f3; r.l3 = l3; r.f10 = f10; r.l10 = l10; r.f20 = f20; r.l20 = l20;

    // order statistics for numpy.percentile's linear interpolation
    const int r0 = prm.rank < 0 ? 0 : (prm.rank > n - 1 ? n - 1 : prm.rank);
    if (!rf_select_hist(x, n, (unsigned)r0, (float)mean, mean - mean == 0.0, sh, r.q0, r.q1, tid))
        rf_select_pair(x, n, (unsigned)r0, sh, r.q0, r.q1, tid);
    rf_finish(r, n, prm, sh, o_stats, o_thr, tid);
    if (o_idx && o_cnt) rf_peaks(x, n, prm, sh, o_idx, o_cnt, tid);
}

// Rows of up to 16 values per thread (n <= 4096: the reference's frame length and below): the row is read once
// more into registers, and nothing after that reads it again except the neighbour compares of the peak scan.
// TIMING-ONLY knock-outs for tools/f1_knockout.sh (results are WRONG with any bit set; the shipped build has 0): what a phase
// costs = the time that disappears when it does.  1 = scan A's 10^(x/10) and float64 sums, 2 = scan B's float64 moments,
// 4 = the histogram select (atomics, prefix scan, candidate rank), 8 = the band edges, 32 = the peak scan's exit-state tables,
// 64 = its walk and replay (wave 0).  ("peaks off" — candidates too — is d_idx = NULL at run time.)
#ifndef SDRK_F1_SKIP
#define SDRK_F1_SKIP 0
#endif

template <class RowPtr>
__device__ __forceinline__ void rf_small(RowPtr x, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                         double* __restrict__ o_stats, double* __restrict__ o_thr,
                                         int* __restrict__ o_idx, int* __restrict__ o_cnt, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RowFeatValues r;
    float xv[16];
    SDRK_PHASE("row_pickup_zero_hist");
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = tid + RF_THREADS * j;
        xv[j] = i < n ? x[i] : 0.0f;
    }
    {   // histogram of the select: thread t owns bins 8t .. 8t+7 here and in the prefix scan
        typedef unsigned rf_v4u __attribute__((ext_vector_type(4)));
        rf_v4u* b4 = reinterpret_cast<rf_v4u*>(sh.bins);
        unsigned zero = 0u;                                    // (made here: a constant zero vector gets hoisted out of the
        asm volatile("" : "+v"(zero));                         //  caller's row loop into four registers and then spilled)
        const rf_v4u z = {zero, zero, zero, zero};
        b4[2 * tid] = z;
        b4[2 * tid + 1] = z;
        if (tid == 0) sh.sel[3] = 0;
    }

    // scan A: max (+ first argmax), sum of x, sum of p = max(10^(x/10), 1e-15), sum of the unclipped x (for mean ln p).
    // Bins under -150 dB are clipped by np.clip(p, 1e-15, None); a wave that holds none (every wave of an ordinary row)
    // takes the loop without the clip selects: su = sum of all its x, sc = 0.
    SDRK_PHASE("scanA_max_sums_pow10");
    float mx = -INFINITY;
    int amx = 0x7fffffff, nclip = 0;
    double su = 0.0, sc = 0.0, sp = 0.0;      // unclipped x, clipped x, p
    float vmin = INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) vmin = fminf(vmin, tid + RF_THREADS * j < n ? xv[j] : INFINITY);
    const bool careful = __any(!(vmin > -149.0f));
    if (!careful) {
        SDRK_PHASE("scanA_sums_ordinary_row");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = tid + RF_THREADS * j;
            if (i < n) {
                const float v = xv[j];
                mx = fmaxf(mx, v);
                if (!(SDRK_F1_SKIP & 1)) {
                    su += (double)v;
                    sp += (double)rf_pow10_tenth_f32_inrange(v);
                }
            }
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // four elements in flight at a time (register pressure)
        }
    } else {
        SDRK_PHASE("scanA_sums_RARE_bins_under_minus_150dB");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = tid + RF_THREADS * j;
            if (i < n) {
                const float v = xv[j];
                mx = fmaxf(mx, v);
                const double dv = (double)v;
                const float p = rf_pow10_tenth_f32(v);
                const bool clip = p < 1e-15f;                 // np.clip(p, 1e-15, None) (NaN stays NaN)
                sp += clip ? 1e-15 : (double)p;
                su += clip ? 0.0 : dv;
                sc += clip ? dv : 0.0;
                nclip += clip ? 1 : 0;
            }
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Values past 300 dB (|X| > 1e15): float32 overflows 10^(x/10) at 385 dB where the reference's float64 does not.
    // Such a wave (none of any real row) sums p once more, relative to its own maximum, and scales the sum back in
    // float64 below.
    SDRK_PHASE("scanA_level_check");
    float level = 0.0f;
    if (__any(mx > 300.0f)) {
        SDRK_PHASE("scanA_RARE_values_over_300dB");
        level = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rf_wave_scan_max(mx)), 63));
        sp = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v = xv[j];
            if (tid + RF_THREADS * j < n && !(v < -150.0f)) sp += (double)rf_pow10_tenth_f32(v - level);
        }
    }
    SDRK_PHASE("scanA_wave_reduce_argmax");
    double sx = su + sc;
    // the wave's maximum (NaNs skipped, as a running `v > mx` does), then its first bin: bin i = 256 j + 64 wave +
    // lane, so the first set bit of the first non-empty ballot over j.  (A wave whose maximum stayed -inf has no
    // bin ABOVE -inf: no argmax, as before.)
    mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rf_wave_scan_max(mx)), 63));
    if (mx > -INFINITY) {
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            const unsigned long long b = __ballot(tid + RF_THREADS * j < n && xv[j] == mx);
            if (b) amx = RF_THREADS * j + 64 * wave + __builtin_ctzll(b);
        }
    }
    sx = rf_wave_scan_add(sx);
    sp = rf_wave_scan_add(sp);
    if (careful) {
        su = rf_wave_scan_add(su);
        nclip = (int)rf_wave_scan_add((unsigned)nclip);
    } else {
        su = sx;
    }
    if (level != 0.0f) sp = sp * rf_pow10_tenth_f64(level) + (double)nclip * 1e-15;   // (the clipped bins' 1e-15 each)
    if (lane == 63) {
        sh.f[wave] = mx; sh.i[wave] = amx; sh.i[28 + wave] = nclip;
        sh.d[wave * 3] = sx; sh.d[wave * 3 + 1] = sp; sh.d[wave * 3 + 2] = su;
    }
    __syncthreads();     // partials of scan A; the zeroed histogram
    SDRK_PHASE("scanA_merge");
    rf_opaque(xv);       // (keeps the compiler from carrying scan A's float64 copies of the row into scan B)
    mx = sh.f[0]; amx = sh.i[0];
    for (int w = 1; w < 4; ++w)
        if (sh.f[w] > mx || (sh.f[w] == mx && sh.i[w] < amx)) { mx = sh.f[w]; amx = sh.i[w]; }
    const double sum_x = sh.d[0] + sh.d[3] + sh.d[6] + sh.d[9];
    const double mean = sum_x / n;
    r.mean_p = (sh.d[1] + sh.d[4] + sh.d[7] + sh.d[10]) / n;
    // mean ln p: ln(10^(x/10)) = x ln(10)/10 for the unclipped bins, ln(1e-15) for the clipped ones
    r.mean_lp = ((sh.d[2] + sh.d[5] + sh.d[8] + sh.d[11]) * 0.23025850929940456840 +
                 (double)(sh.i[28] + sh.i[29] + sh.i[30] + sh.i[31]) * -34.538776394910684) / n;

    // scan B: central moments; occupied-band edges (thresholds in float32, as peak - float(drop) is) from wave
    // ballots — bin i = 256 j + 64 wave + lane, so the first / last set bit of the first / last non-empty ballot
    // is the wave's first / last index, kept in scalar registers; the histogram of the select
    SDRK_PHASE("scanB_moments_edges_histogram");
    const float t3 = mx - 3.0f, t10 = mx - 10.0f, t20 = mx - 20.0f;
    // histogram coordinate t(v) = 16 v - 16 (mean - 64): one fma, monotone in v; bin = floor(clamp(t, 0, 2047))
    const float hoff = -RF_BINS_PER_DB * ((float)mean - (float)(RF_BINS / 2) / RF_BINS_PER_DB);
    auto coord = [&](float v) { return fmaf(v, RF_BINS_PER_DB, hoff); };
    auto bin_of = [&](float v) { return (int)fminf(fmaxf(coord(v), 0.0f), (float)(RF_BINS - 1)); };
    double s2 = 0.0, s4 = 0.0;
    int f3 = 0x7fffffff, l3 = -1, f10 = 0x7fffffff, l10 = -1, f20 = 0x7fffffff, l20 = -1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = tid + RF_THREADS * j;
        const float v = xv[j];
        if (i < n) {
            // (d2 as a product of its own, never contracted into `s2 += dv * dv`: the compiler's choice differed between the
            //  fused and the stand-alone build of this very loop once the code around it changed — the last ulp of the
            //  variance in 4 % of the rows; the fourth moment's fma is written out)
            if (!(SDRK_F1_SKIP & 2)) {
                const double dv = (double)v - mean, d2 = rf_mul_f64(dv, dv);
                s2 += d2;
                s4 = fma(d2, d2, s4);
            }
            if (!(SDRK_F1_SKIP & 4)) atomicAdd(&sh.bins[bin_of(v)], 1u);
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // Band edges: only the wave's FIRST and LAST bin at or above each threshold are wanted, and bin i = 256 j + 64 wave +
    // lane grows with j — so look for the first non-empty ballot from j = 0 up and for the last from j = 15 down, and
    // stop there (wave-uniform exits).  An ordinary row has bins within 20 dB of its peak in every 64-bin stretch: two
    // compares per threshold instead of sixteen (round 4 took every ballot of every j: 48 compares and ~26 scalar
    // instructions per j).  The sets are nested (within 3 dB => within 10 => within 20): a wave without a bin within
    // 20 dB is done after its first sixteen compares.
    SDRK_PHASE("scanB_band_edges_first_last_search");
    auto edges = [&](float t, int& first, int& last) -> bool {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned long long b = __ballot(tid + RF_THREADS * j < n && xv[j] >= t);
            if (b) { first = RF_THREADS * j + 64 * wave + __builtin_ctzll(b); break; }
        }
        if (first == 0x7fffffff) return false;
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            const unsigned long long b = __ballot(tid + RF_THREADS * j < n && xv[j] >= t);
            if (b) { last = RF_THREADS * j + 64 * wave + 63 - __builtin_clzll(b); break; }
        }
        return true;
    };
    if (!(SDRK_F1_SKIP & 8) && edges(t20, f20, l20) && edges(t10, f10, l10)) edges(t3, f3, l3);
    SDRK_PHASE("scanB_wave_reduce_merge");
    s2 = rf_wave_scan_add(s2);
    s4 = rf_wave_scan_add(s4);
    if (lane == 63) {
        int* s = sh.i + 4 + wave * 6;
        s[0] = f3; s[1] = l3; s[2] = f10; s[3] = l10; s[4] = f20; s[5] = l20;
        sh.d2[wave * 2] = s2; sh.d2[wave * 2 + 1] = s4;
    }
    __syncthreads();     // partials of scan B; the histogram
    rf_opaque(xv);       // (... nor scan B's bin numbers into the candidate pass)
    for (int w = 0; w < 4; ++w) {
        const int* s = sh.i + 4 + w * 6;
        f3 = min(f3, s[0]); l3 = max(l3, s[1]); f10 = min(f10, s[2]); l10 = max(l10, s[3]);
        f20 = min(f20, s[4]); l20 = max(l20, s[5]);
    }
    r.m2 = (sh.d2[0] + sh.d2[2] + sh.d2[4] + sh.d2[6]) / n;
    r.m4 = (sh.d2[1] + sh.d2[3] + sh.d2[5] + sh.d2[7]) / n;
    r.mx = mx; r.amx = amx; r.mean = mean;
    r.f3 = f3; r.l3 = l3; r.f10 = f10; r.l10 = l10; r.f20 = f20; r.l20 = l20;

    // order statistics sorted[r0], sorted[r1] (ascending) for numpy.percentile's linear interpolation
    if (SDRK_F1_SKIP & 4) {
        r.q0 = r.q1 = mx;
    } else {
    SDRK_PHASE("select_prefix_scan");
    const unsigned r0 = (unsigned)(prm.rank < 0 ? 0 : (prm.rank > n - 1 ? n - 1 : prm.rank));
    const unsigned r1 = r0 + 1 < (unsigned)n ? r0 + 1 : r0;
    unsigned c[8];
    {
        typedef unsigned rf_v4u __attribute__((ext_vector_type(4)));
        const rf_v4u* b4 = reinterpret_cast<const rf_v4u*>(sh.bins);
        const rf_v4u a = b4[2 * tid], b = b4[2 * tid + 1];
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    }
    const unsigned tot = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
    unsigned incl = rf_wave_scan_add(tot);
    if (lane == 63) sh.wtot[wave] = incl;
    __syncthreads();
    incl += rf_waves_before(sh.wtot, wave);
    const unsigned excl = incl - tot;
    if (excl <= r0 && r0 < incl) {            // exactly one thread: its bins hold rank r0
        unsigned cum = excl;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (r0 >= cum && r0 < cum + c[k]) { sh.sel[0] = 8 * tid + k; sh.sel[2] = (int)cum; }
            cum += c[k];
        }
    }
    if (excl <= r1 && r1 < incl) {
        unsigned cum = excl;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (r1 >= cum && r1 < cum + c[k]) sh.sel[1] = 8 * tid + k;
            cum += c[k];
        }
    }
    __syncthreads();
    const int b0 = __builtin_amdgcn_readfirstlane(sh.sel[0]), b1 = __builtin_amdgcn_readfirstlane(sh.sel[1]);
    const unsigned below = (unsigned)__builtin_amdgcn_readfirstlane(sh.sel[2]);
    // the histogram path needs finite values (NaN / inf have no bin order) and target bins inside the window
    const double finite_probe = sum_x - sum_x;                           // 0 unless a value was NaN or +-inf
    bool fast = __builtin_amdgcn_readfirstlane((int)(finite_probe == 0.0)) && b0 >= 1 && b1 <= RF_BINS - 2;
    SDRK_PHASE("select_candidates_rank");
    if (fast) {
        // bins b0 .. b1 (1 <= b0 <= b1 <= 2046; the bins between them are empty: ranks r0, r1 are adjacent) hold
        // exactly the values with b0 <= t(v) < b1 + 1
        const float t_lo = (float)b0, t_hi = (float)(b1 + 1);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = tid + RF_THREADS * j;
            if (i < n) {
                const float t = coord(xv[j]);
                if (t >= t_lo && t < t_hi) {
                    const int pos = atomicAdd(&sh.sel[3], 1);
                    if (pos < RF_CAND) sh.cand[pos] = xv[j];
                }
            }
        }
        __syncthreads();
        const int K = __builtin_amdgcn_readfirstlane(sh.sel[3]);
        fast = K <= RF_CAND;
        if (fast && wave == 0) {
            // ascending rank inside the candidate set (ties by list position), by every lane against every candidate
            const float mine = lane < K ? sh.cand[lane] : INFINITY;
            unsigned rank = 0;
            for (int j = 0; j < K; ++j) {
                const float cj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), j));
                rank += (cj < mine || (cj == mine && j < lane)) ? 1u : 0u;
            }
            const unsigned g = below + rank;
            const unsigned long long m0 = __ballot(lane < K && g == r0), m1 = __ballot(lane < K && g == r1);
            r.q0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), (int)__builtin_ctzll(m0 | (1ull << 63))));
            r.q1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), (int)__builtin_ctzll(m1 | (1ull << 63))));
        }
    }
    SDRK_PHASE("select_fallback_radix");
    if (!fast) rf_select_pair(x, n, r0, sh, r.q0, r.q1, tid);
    }
    SDRK_PHASE("finish_threshold_stats_out");
    rf_finish(r, n, prm, sh, o_stats, o_thr, tid);      // (fast path: thread 0 is in wave 0, which holds q0 / q1)
    if (o_idx && o_cnt) {
        if (prm.min_distance >= 1 && prm.min_distance <= 16) rf_peaks_small(xv, x, n, prm, sh, o_idx, o_cnt, tid);
        else rf_peaks(x, n, prm, sh, o_idx, o_cnt, tid);
    }
}

// ---- greedy peak scan of a row of up to 4096 bins, min_distance <= 16 ----------------------------------------------
// classifier.py:200-212 accepts candidates left to right when they lie >= d bins after the previously accepted one: a
// serial recurrence, one dependent scalar chain per accepted peak (~70 clocks each; noise rows accept ~250).  Its
// state at a word boundary is only "how many leading bins of the next 64-bin word are still suppressed" — s in
// [0, d).  So:  (1) all four waves evaluate every word for EVERY entry state (lane = 16 word' + s, four words per
// wave instruction) and record the exit state: a 16-nibble table per word;  (2) one wave walks the 64 tables (one
// scalar lookup per word) and learns every word's true entry state;  (3) lane w replays word w from that state,
// which yields its accepted bins, and the indices leave in order by a prefix sum of the counts.  Exact by
// construction (the same recurrence, evaluated speculatively); ~4x less time on the critical wave.
// One 64-bin word: candidates (mlo, mhi) with the entry suppression already applied; accepts greedily, low half then
// high half, all in 32-bit operations (d <= 16 < 32: a span reaches at most into the next half).  `last` = the last
// accepted bin of the word or -1; (alo, ahi) = the accepted bins when ACC.  Trip counts are wave-uniform: the lane
// with the most accepted peaks in a half.
template <bool ACC>
__device__ __forceinline__ void rf_accept_word(unsigned mlo, unsigned mhi, int d, unsigned& alo, unsigned& ahi, int& last) {
    const unsigned dspan = (1u << d) - 1u;
    alo = 0; ahi = 0; last = -1;
    while (__any(mlo != 0)) {
        if (mlo != 0) {
            const int bit = __builtin_ctz(mlo);
            if (ACC) alo |= 1u << bit;
            last = bit;
            mlo &= ~(dspan << bit);                       // bins past 31 fall off: handled below
        }
    }
    const int carry = last + d - 32;                       // leading bins of the high half still suppressed (< 16)
    mhi = carry > 0 ? mhi & (~0u << carry) : mhi;
    while (__any(mhi != 0)) {
        if (mhi != 0) {
            const int bit = __builtin_ctz(mhi);
            if (ACC) ahi |= 1u << bit;
            last = 32 + bit;
            mhi &= ~(dspan << bit);
        }
    }
}

// The same for spacings d >= 11 — the reference's max(3, N // 300) from N = 3300 up, i.e. at its own frame length: a 32-bin
// half then holds at most three accepted peaks (bins b, b + d, b + 2 d <= 31), so the two data-dependent loops become 3 + 3
// straight-line steps of four instructions — no wave-wide `any`, no branch, no exec-mask juggling.  An exhausted mask stays
// zero through the remaining steps: ffs(0) - 1 = -1 never raises `last`, and the span shifted by it clears nothing.
template <bool ACC>
__device__ __forceinline__ void rf_accept_word3(unsigned mlo, unsigned mhi, int d, unsigned& alo, unsigned& ahi, int& last) {
    const unsigned dspan = (1u << d) - 1u;
    alo = 0; ahi = 0;
    int llo = -1, lhi = -1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int bit = __builtin_ffs((int)mlo) - 1;
        if (ACC) alo |= mlo & (0u - mlo);
        llo = max(llo, bit);
        mlo &= ~(dspan << (bit & 31));
    }
    const int carry = max(llo + d - 32, 0);                // leading bins of the high half still suppressed (< 16)
    mhi &= ~0u << carry;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int bit = __builtin_ffs((int)mhi) - 1;
        if (ACC) ahi |= mhi & (0u - mhi);
        lhi = max(lhi, bit);
        mhi &= ~(dspan << (bit & 31));
    }
    last = lhi >= 0 ? 32 + lhi : llo;
}

// wave-uniform choice between the two (d is a kernel argument)
template <bool ACC>
__device__ __forceinline__ void rf_accept(unsigned mlo, unsigned mhi, int d, unsigned& alo, unsigned& ahi, int& last) {
    if (d >= 11) rf_accept_word3<ACC>(mlo, mhi, d, alo, ahi, last);
    else rf_accept_word<ACC>(mlo, mhi, d, alo, ahi, last);
}

template <class RowPtr>
__device__ __forceinline__ void rf_peaks_small(const float (&xv)[16], RowPtr x, int n, const RowFeatParams& prm,
                                               RowFeatShared& sh, int* __restrict__ o_idx, int* __restrict__ o_cnt, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = prm.min_distance;
    // v > thr (float64 threshold) <=> v > the largest float32 <= thr: one float compare per bin
    SDRK_PHASE("peaks_candidates_ballots");
    float thr_f = (float)sh.thr;
    if ((double)thr_f > sh.thr) thr_f = nextafterf(thr_f, -INFINITY);
    // candidates: strict local maxima above the threshold; word 4 j + wave holds bins 256 j + 64 wave + lane
    // Both neighbours of every bin are fetched unconditionally, four bins at a time, and the three compares are combined
    // without short-circuit.  (Written as `v > thr && v > x[i - 1] && v > x[i + 1]` the compiler made each `&&` a branch with
    // its own row read and `s_waitcnt lgkmcnt(0)` behind it: thirty-two exposed LDS latencies per thread, 2.3 ns of the
    // 21.6 ns per row — round 5's knock-out measurement.)  Indices are clamped into the row; the ends are masked out below.
    unsigned flo = 0, fhi = 0;          // lane j collects the ballot of bins 256 j + 64 wave ..., written once below
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
        float lft[4], rgt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + RF_THREADS * (j0 + q);
            lft[q] = x[min(max(i - 1, 0), n - 1)];
            rgt[q] = x[min(i + 1, n - 1)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q, i = tid + RF_THREADS * j;
            const float v = xv[j];
            const bool cand = (i >= 1) & (i < n - 1) & (v > thr_f) & (v > lft[q]) & (v > rgt[q]);
            const unsigned long long b = __ballot(cand);
            flo = lane == j ? (unsigned)b : flo;          // (v_writelane through inline assembly saved a compare per ballot and lost a
            fhi = lane == j ? (unsigned)(b >> 32) : fhi;  //  ballot in about one row of 5 000: no builtin for it in this clang, so selects)
        }
    }
    if (lane < 16) sh.flags[4 * lane + wave] = ((unsigned long long)fhi << 32) | flo;
    __syncthreads();
    // (1) exit-state tables: wave k owns words 16 k .. 16 k + 15
    SDRK_PHASE("peaks_exit_state_tables");
    const int st = lane & 15, grp = lane >> 4;
#pragma unroll 1
    for (int round = 0; round < ((SDRK_F1_SKIP & 32) ? 0 : 4); ++round) {
        const int w = 16 * wave + 4 * round + grp;
        const unsigned long long m = sh.flags[w];
        unsigned alo, ahi;
        int last;
        rf_accept<false>((unsigned)m & (~0u << st), (unsigned)(m >> 32), d, alo, ahi, last);
        // exit state: bins of the next word still inside the last accepted peak's span
        int ex = last + d - 64;
        ex = ex < 0 ? 0 : ex;
        unsigned lo = st < 8 ? (unsigned)ex << (4 * st) : 0u, hi = st >= 8 ? (unsigned)ex << (4 * (st - 8)) : 0u;
        // OR over the 16 entry states of the word (one DPP row): lane 15 of the row ends up with the table
#define RF_STEP(CTRL)                                                                      \
        lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, false);    \
        hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, false);
        RF_STEP(0x111) RF_STEP(0x112) RF_STEP(0x114) RF_STEP(0x118)
#undef RF_STEP
        if (st == 15) sh.tbl[w] = ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    if (wave == 0 && !(SDRK_F1_SKIP & 64)) {
        // (2) the walk: entry state of every word
        SDRK_PHASE("peaks_walk_wave0");
        // Lane w needs the state in which word w is entered = table(w - 1)[entry(w - 1)], entry(0) = 0: a chain through all 64
        // words.  But the greedy chains from different entry states merge quickly, so most tables are CONSTANT over the entry
        // states that can occur (0 .. d - 1) — and a word behind a constant table knows its entry state without the chain.  Those
        // resolve at once; the others wait for their predecessor, one hop per round (64 rounds in the worst case, one or two on
        // real rows).  (Round 4 walked the tables serially on the scalar unit: 64 x 7 dependent instructions, ~2 000 cycles of one
        // wave per row.)
        const unsigned long long Tprev = lane ? sh.tbl[lane - 1] : 0ull;         // ("before word 0": a constant 0)
        const unsigned long long maskd = d >= 16 ? ~0ull : ((1ull << (4 * d)) - 1ull);
        const unsigned c0 = (unsigned)Tprev & 15u;
        bool known = ((Tprev ^ (0x1111111111111111ull * c0)) & maskd) == 0ull;
        int entry = (int)c0;
        while (__any(!known)) {
            const int pe = __shfl_up(entry, 1, 64), pk = __shfl_up((int)known, 1, 64);
            if (!known && pk) {
                entry = (int)((unsigned)(Tprev >> (4 * pe)) & 15u);
                known = true;
            }
        }
        // (3) replay word `lane` from its entry state; indices out in order
        SDRK_PHASE("peaks_replay_emit_wave0");
        const unsigned long long m = sh.flags[lane];
        unsigned alo, ahi;
        int last;
        rf_accept<true>((unsigned)m & (~0u << entry), (unsigned)(m >> 32), d, alo, ahi, last);
        const unsigned pc = (unsigned)(__popc(alo) + __popc(ahi));
        const unsigned incl = rf_wave_scan_add(pc);
        unsigned pos = incl - pc;
        while (__any(alo != 0)) {
            if (alo != 0) {
                const int bit = __builtin_ctz(alo);
                alo &= alo - 1;
                if ((int)pos < prm.max_peaks) o_idx[pos] = 64 * lane + bit;
                ++pos;
            }
        }
        while (__any(ahi != 0)) {
            if (ahi != 0) {
                const int bit = __builtin_ctz(ahi);
                ahi &= ahi - 1;
                if ((int)pos < prm.max_peaks) o_idx[pos] = 64 * lane + 32 + bit;
                ++pos;
            }
        }
        if (lane == 63) *o_cnt = (int)incl;
    }
}

// The whole measurement of one row by one 256-thread workgroup.  `x` points at n float32 values (LDS or global).
// o_stats: 16 doubles; o_thr: 1 double; o_idx: max_peaks ints; o_cnt: 1 int (total peaks found, may exceed max_peaks).
// o_idx / o_cnt / o_thr may be null (stats only).
// SIZE: 0 = either routine by n (the fused kernel passes a constant n and keeps one), 1 = the caller guarantees
// n <= 16 * RF_THREADS, 2 = the caller guarantees n > 16 * RF_THREADS (row_features.hip builds one kernel per class,
// so that the short-row kernel is not allocated registers for code it never runs).
template <int SIZE = 0, class RowPtr>
__device__ __forceinline__ void row_features_wg(RowPtr x, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                                double* __restrict__ o_stats, double* __restrict__ o_thr,
                                                int* __restrict__ o_idx, int* __restrict__ o_cnt) {
    // the thread number is re-read "opaquely" per row: inside a persistent row loop the compiler would otherwise
    // hoist every address and lane mask derived from it out of the loop and hold ~100 registers for them
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < RF_THREADS);   // (the range survives the opaque copy: bounds checks against a constant n fold)
    if (SIZE == 1 || (SIZE == 0 && n <= 16 * RF_THREADS)) rf_small(x, n, prm, sh, o_stats, o_thr, o_idx, o_cnt, tid);
    else rf_large(x, n, prm, sh, o_stats, o_thr, o_idx, o_cnt, tid);
}

template <class RowPtr>
__device__ __forceinline__ void rf_peaks(RowPtr x, int n, const RowFeatParams& prm, RowFeatShared& sh,
                                         int* __restrict__ o_idx, int* __restrict__ o_cnt, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    // greedy peak scan.  All four waves mark the candidates of a 4096-bin stretch (64 ballot words in LDS), then wave 0
    // applies the spacing rule in index order: two barriers per stretch instead of two per 256 bins.
    const double thr = sh.thr;
    for (int sbase = 0; sbase < n; sbase += 4096) {
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            const int i = sbase + 256 * c + tid;
            // all three reads unconditionally (indices clamped into the row, the ends masked): as `… && v > x[i - 1] && v > x[i + 1]`
            // each `&&` became a branch with its own read and a full wait behind it (see rf_peaks_small)
            const float v = x[min(i, n - 1)], lft = x[min(max(i - 1, 0), n - 1)], rgt = x[min(i + 1, n - 1)];
            const bool cand = (i >= 1) & (i < n - 1) & ((double)v > thr) & (v > lft) & (v > rgt);
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
