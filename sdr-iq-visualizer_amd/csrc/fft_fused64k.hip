// fft_fused64k.hip — N = 65536 (BASELINE.json config 3: the waterfall STFT) in ONE persistent launch whose
// intermediate never leaves the XCD it was produced on.
//
// Same arithmetic as the two launches of fft_tiled2.hip for 256 x 256 (col pass: DFT-256 down 16-wide column
// tiles, times W_N^(m k3); row pass: DFT-256 along 16 adjacent rows, fftshift, log epilogue) and the same
// code for both (fft_lds_core.h) — bit-identical results — but the 512 KiB complex64 intermediate of a frame
// lives in a one-slot ring inside ONE XCD's 4 MiB L2 instead of crossing the fabric twice.
//
// Roles.  The workgroups of the persistent grid (3 per CU, 256 threads) organise themselves into SETS of 32
// that share an L2: every workgroup reads its XCC id (s_getreg HW_REG_XCC_ID — HIP promises nothing about
// placement, so nothing is assumed), draws a ticket r from that XCD's counter and becomes member r % 32 of the
// XCD's set r / 32: members 0..15 are the col workgroups of tile positions 0..15, members 16..31 the row
// workgroups of row tiles 0..15.  Member 0 gives the set a dense number g (global counter), which selects the
// contiguous run of frames the set owns and its ring.  A role never changes, so a col workgroup keeps its
// window coefficients and W_N^(m k3) factors in registers, prefetches the next frame's samples across the
// current transform and — for the overlapped frames of an STFT — re-uses the samples two consecutive frames
// share (the software pipeline of col_pass_kernel<.., FIXED, SH>).
//
// Hand-over inside a set (frame number s of the run, ring slot s % D; D = 1).  The ring is laid out in 16 x 16 blocks
// (scratch_index): block (k3 tile q, column tile c) is written by col workgroup c ALONE — its q-th store — and read by
// row workgroup q ALONE — its c-th load.  So the dependencies are point to point and the hand-over is PROGRESSIVE:
//   col_done[c][wave 0..3] = s + 1   once that wave's stores of frame s are in the L2
//   row_done[q][wave 0..3] = s + 1   once that wave's loads of frame s have returned
//   a row wave issues its load c as soon as col_done[c][*] > s; a col wave issues its store q as soon as
//   row_done[q][*] > s - D.  Every wave polls for itself (one 64-lane load covers the partner role's 64 words) and
//   issues what has become possible, in whatever order the partners finish; nobody waits for all sixteen partners
//   before touching the first block, and there is no workgroup barrier in the hand-over.
// Waits only point backwards in s, so there is no cycle; every spin is bounded, and a timeout (a set that never
// became complete because some of its members were not resident) raises the error word instead of hanging — the
// host then reports the launch as failed.
//
// Visibility inside one XCD: a producer's plain stores are complete in the shared L2 once its `s_waitcnt vmcnt(0)`
// returns; it then publishes with a PLAIN store (st_flag below: the flag stays in that L2).  The consumer polls with
// agent-scope (sc1) loads and reads the ring with sc1 loads, which bypass its CU's L1 and are served by that same
// L2.  No L2 write-back is needed because producer and consumer were verified, by XCC id, to share the L2.
//
// Measured (MI355X, round 6, profiles/r06/fused64k_policy.md): 4096 packed Hann frames 1.04 ms against 1.26 ms for the two
// tiled launches; BASELINE config 3 (18 749 frames, hop 32768) 3.83-3.94 ms against 4.98-5.04.  Rounds 1-5 had it at 2.2-2.4 ms /
// 8.7 ms: the flags were agent-scope stores, which write through and DROP the line from the L2, so every poll went to the
// fabric.  What the ring costs the fabric depends on how much of the L2 it takes (WRITE_SIZE per 4096 packed frames, 1.07 GB
// of rows: 3 sets x 2 slots 3.38 GB, 3 x 1 1.50 GB, 1 x 2 with write-through row stores 1.10 GB) — but the kernel is not bound
// there: with all three sets of an XCD aliased onto one 1 MiB ring and every wait compiled out (traffic exactly 1.0 x the
// algorithmic bytes) it still takes 1.08 ms.  What bounds it is the cycle of a set: col store -> flag -> row load -> flag,
// about 5 us per frame against 2.8 us of work per workgroup and frame (phase trace, same file), with three frames in flight per XCD.
#include "fft_lds_core.h"

namespace sdrk {

constexpr int FU_THREADS = 256;
constexpr int FU_N = 65536;
constexpr int FU_A = 256;                  // = M
#ifndef FU_RING_SLOTS_N
#define FU_RING_SLOTS_N 1                  // depth 2 costs more L2 (write-backs of dead ring lines) than the overlap it allows gains
#endif
// Experiment switches (experiments/fused64k_policy/, never defined in the product build; all need -DSDRK_FUSED_EXPERIMENT):
//   FU_KNOCK     timing only, wrong results; bits: 1 every workgroup a col workgroup, 2 every one a row workgroup, 4 no ring
//                traffic, 8 no transforms, 16 no input loads, 32 no output stores
//   FU_ALIAS_RING  timing only (with NOWAIT): every set of an XCD uses the same ring
//   FU_TRACE     100 MHz timestamps of the phases of every workgroup of sets 0 and 7, first 96 frames, written behind the
//                output rows (the caller allocates 1 MiB more): [set sel 2][member 32][frame 96][slot 8] unsigned
#if !defined(SDRK_FUSED_EXPERIMENT) || !defined(FU_KNOCK)
#undef FU_KNOCK
#define FU_KNOCK 0
#endif
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_TRACE)
#define FU_T(slot)                                                                                                    \
    do {                                                                                                              \
        if (tr_on && tid == 0 && s < 96u)                                                                             \
            tr_base[((size_t)tr_sel * 96 + s) * 8 + (slot)] = (unsigned)__builtin_amdgcn_s_memrealtime();              \
    } while (0)
#else
#define FU_T(slot) do { } while (0)
#endif
#ifndef FU_POLL_SLEEP
#define FU_POLL_SLEEP 1                    // x 64 clocks between two polls that made no progress (2 / 4 / 8: within noise, r06)
#endif
constexpr int FU_RING_SLOTS = FU_RING_SLOTS_N;   // D; x 512 KiB per set, three sets per XCD
constexpr int FU_MAX_XCD = 16;
constexpr int FU_MAX_SETS = 64;            // dense set numbers (grid / 32 <= this)
constexpr unsigned FU_SPIN_LIMIT = 1u << 20;  // polls of >= 0.2 us each: give up after a fraction of a second

// control block (unsigned words), zeroed before every launch:
//   [0] dense set counter   [1] error flag   [2..7] record of the first timeout
//   [32 + x]                       ticket counter of XCD x
//   [64 + x * 8 + j]               dense number + 1 of set j of XCD x (published by its member 0)
//   [256 + g * 128 ...]            set g: col_done[64] then row_done[64]
constexpr size_t FU_CTRL_WORDS = 256 + (size_t)FU_MAX_SETS * 128;

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A set's own flags (col_done / row_done) are stored WITHOUT scope bits.  An agent-scope store is `global_store_dword ... sc1`,
// which writes through and DROPS the line from the L2 (MI355X_MICROARCH.md, "stores of each flavour"): every poll then missed
// the L2 and went to the fabric — rounds 1-5 paid 1.2 ms per 4096 frames for that, more than the transform itself (2.41 -> 1.20 ms,
// profiles/r06/fused64k_policy.md).  A plain store stays in the L2 that both sides were verified, by XCC id, to share; the
// pollers' sc1 loads bypass their CU's L1 and are served by that L2.  Ordering: the storing wave has waited `vmcnt(0)` for its
// ring accesses before this store is issued, and the asm's memory clobber keeps the compiler from moving anything across it.
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// One poll of the partner role's 64 flag words ([workgroup c][wave 0..3]).  Bit 4c of the result: all four waves of partner
// workgroup c have published >= want.  Wave-uniform.
__device__ __forceinline__ unsigned long long ready_groups(const unsigned* words, unsigned want) {
    const unsigned f = ld_agent(words + (threadIdx.x & 63));
    unsigned long long ok = __builtin_amdgcn_ballot_w64(f >= want);
    ok &= ok >> 1;
    ok &= ok >> 2;
    asm volatile("" ::: "memory");   // no access of the ring may be hoisted above the poll that allows it
    return ok & 0x1111111111111111ull;
}
// Between two polls that made no progress.  false: give up (timeout, or somebody else raised the error flag); wave-uniform.
__device__ __forceinline__ bool prog_spin(unsigned& spins, bool progressed, const unsigned* words, unsigned want, unsigned* ctrl,
                                          unsigned site) {
    if (progressed) return true;
    if ((++spins & 31u) == 0u && (spins > FU_SPIN_LIMIT || ld_agent(ctrl + 1))) {
        if ((threadIdx.x & 63) == 0 && !ld_agent(ctrl + 1)) {   // first reporter leaves a record
            ctrl[2] = (unsigned)(words - ctrl); ctrl[3] = spins; ctrl[4] = want; ctrl[5] = site;
            st_agent(ctrl + 1, 1u);
        }
        return false;
    }
    __builtin_amdgcn_s_sleep(FU_POLL_SLEEP);
    return true;
}

// IN_AUX / OUT_AUX: cache-policy bits of the streamed input loads and row stores (buffer aux: 1 = sc0, 2 = nt, 16 = sc1); NOWAIT
// compiles the hand-over's waits out (wrong results; for timing the traffic alone).  The product instantiates the defaults only;
// experiments/fused64k_policy (round 6) instantiates the sweep behind -DSDRK_FUSED_EXPERIMENT.
template <bool HAS_WINDOW, int EPILOGUE, int SH, int IN_AUX = 2, int OUT_AUX = 2, bool NOWAIT = false>
__global__ __launch_bounds__(FU_THREADS, 3) void fused64k_kernel(
    const float2* __restrict__ iq, size_t frame_stride, void* __restrict__ out_raw, unsigned n_frames,
    const float* __restrict__ window, const float2* __restrict__ tw256, const float2* __restrict__ t1T,
    const float2* __restrict__ t2, float2* __restrict__ ring, unsigned* __restrict__ ctrl, unsigned n_sets,
    unsigned sets_per_xcd, float eps, int shift) {
    using C = LdsCfg<8>;
    constexpr int M = FU_A, T = C::T, D = FU_RING_SLOTS;   // T = 16 threads per 256-point transform
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];   // 16 x SLOT (exchange) / [256][17] transpose tile
    __shared__ unsigned sh_role[4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- role ----
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= FU_MAX_XCD - 1;
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_SEGREGATE)
        // experiment: roles by CU instead of by arrival — the CUs of shader engines 0 and 2 host col workgroups only, those of
        // 1 and 3 row workgroups only (HW_ID: CU bits 8-11, SE bits 13-15, workgroup slot bits 16-19; experiments/probes/
        // hwid_probe.hip: 8 CUs per engine, 3 workgroups per CU), so that the L2-hit ring loads and polls of the row role never
        // queue behind the input stream's HBM misses in their own CU's memory pipeline
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const bool col_cu = ((hw >> 13) & 1u) == 0u;
        const unsigned r16 = __hip_atomic_fetch_add(ctrl + (col_cu ? 32 : 48) + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned set = r16 >> 4, member = (r16 & 15) + (col_cu ? 0u : 16u);
#else
        const unsigned r = __hip_atomic_fetch_add(ctrl + 32 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned set = r >> 5, member = r & 31;
#endif
        unsigned g = 0xFFFFFFFFu;
        if (set < sets_per_xcd && set < 8) {
            unsigned* slot = ctrl + 64 + xcc * 8 + set;
            if (member == 0) {
                g = __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st_agent(slot, g + 1);
            } else {
                unsigned v = ld_agent(slot), spins = 0;
                while (v == 0 && ++spins < FU_SPIN_LIMIT && !ld_agent(ctrl + 1)) {
                    __builtin_amdgcn_s_sleep(2);
                    v = ld_agent(slot);
                }
                if (v == 0) {
                    if (!ld_agent(ctrl + 1)) { ctrl[2] = 64 + xcc * 8 + set; ctrl[5] = 3; }
                    st_agent(ctrl + 1, 1u);
                } else {
                    g = v - 1;
                }
            }
        }
        sh_role[0] = g;
        sh_role[1] = member;
        sh_role[2] = xcc;
        sh_role[3] = set;
    }
    __syncthreads();
    const unsigned g = sh_role[0], member = sh_role[1];
    if (g >= n_sets) return;                         // a surplus workgroup, or the set never got its number
    unsigned* __restrict__ col_done = ctrl + 256 + (size_t)g * 128;
    unsigned* __restrict__ row_done = col_done + 64;
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_ALIAS_RING)
    float2* __restrict__ my_ring = ring + (size_t)sh_role[2] * D * FU_N;
#else
    float2* __restrict__ my_ring = ring + (size_t)g * D * FU_N;
#endif
    const size_t run = ((size_t)n_frames + n_sets - 1) / n_sets;
    const size_t f_begin = g * run, f_end = f_begin + run < n_frames ? f_begin + run : n_frames;
    if (f_begin >= f_end) return;

    const unsigned pos = member & 15;
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_TRACE)
    const bool tr_on = g == 0 || g == 7;
    const int tr_sel = (g == 0 ? 0 : 32) + (int)member;
    unsigned* __restrict__ tr_base = reinterpret_cast<unsigned*>(static_cast<float*>(out_raw) + (size_t)n_frames * FU_N);
#endif
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_ROLE_MIX)
    // experiment: tickets are drawn in dispatch order, which walks the CUs of an XCD — so members 0..15 of EVERY set land on the
    // same sixteen CUs.  Swapping the halves in every other set puts col and row workgroups on every CU.
    const bool first_half_is_col = (sh_role[3] & 1u) == 0u;
#else
    const bool first_half_is_col = true;
#endif
    const bool is_col = (FU_KNOCK & 1) ? true : (FU_KNOCK & 2) ? false : ((member < 16) == first_half_is_col);
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_PRIO)
    // experiment: raise the issue priority of one role's waves (FU_PRIO = 1: col, 2: row)
    if ((FU_PRIO == 1) == is_col) __builtin_amdgcn_s_setprio(2);
#endif
    if (is_col) {
        // ---------------- col workgroup of tile position `pos` ----------------
        const int fr = tid & 15, tau = tid >> 4;
        LdsTw<8> tw;
        lds_tw_init<8>(tw, tw256, tau);
        const int m = (int)pos * 16 + fr;
        float wreg[16];
        if (HAS_WINDOW) {
#pragma unroll
            for (int q = 0; q < 16; ++q) wreg[q] = window[(size_t)(tau + T * q) * M + m];
        }
        cf bw[16];   // W_N^(m k3), k3 = tau + T q
        {
            const float2 b0 = t1T[tau * M + m];
            const cf base = cf{b0.x, b0.y};
            const float4* __restrict__ row = reinterpret_cast<const float4*>(t2 + (size_t)m * 16);
#pragma unroll
            for (int q2 = 0; q2 < 8; ++q2) {
                const float4 w = row[q2];
                bw[2 * q2] = cmul(base, cf{w.x, w.y});
                bw[2 * q2 + 1] = cmul(base, cf{w.z, w.w});
            }
        }
        const int e0 = tau * M + m;
        constexpr int estep = T * M;
        auto issue_from = [&](size_t f, v2f (&x)[16], int q0) {
            const bool live = f < f_end;
            const __amdgpu_buffer_rsrc_t rx = frame_rsrc(iq + (live ? f : f_begin) * frame_stride, live ? (unsigned)(FU_N * 8) : 0u);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q >= q0) {
                    if ((FU_KNOCK & 16) && !(eps < -1e30f)) { x[q] = v2f{(float)f * 1e-3f + (float)tid, (float)q}; continue; }
                    x[q] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, e0 * 8, q * estep * 8, IN_AUX));
                }
        };
        const int so = scratch_index(tau, m, M);
        v2f xa[16], xb[16];
        issue_from(f_begin, xa, 0);
        for (size_t f = f_begin; f < f_end; ++f) {
            const unsigned s = (unsigned)(f - f_begin);
            FU_T(0);
            issue_from(f + 1, xb, 16 - SH);
            cf v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = HAS_WINDOW ? cf{xa[q].x, xa[q].y} * wreg[q] : cf{xa[q].x, xa[q].y};
            if (!(FU_KNOCK & 8)) lds_fft_core<8, 16>(v, lds_all, fr, tau, tw);
            FU_T(1);
            v2f zz[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = cmul(v[rev16(q)], bw[q]);
                zz[q] = v2f{z.x, z.y};
            }
            FU_T(2);
            // store q goes to row workgroup q alone: it may be issued once THAT workgroup has read frame s - D
            const __amdgpu_buffer_rsrc_t ro = frame_rsrc(my_ring + (size_t)(s % D) * FU_N, (unsigned)(FU_N * 8));
            {
                unsigned long long pending = 0x1111111111111111ull;
                unsigned spins = 0;
                while (pending) {
                    const unsigned long long go =
                        (NOWAIT || s < (unsigned)D) ? pending : (ready_groups(row_done, s - D + 1) & pending);
                    if (!(FU_KNOCK & 4) || eps < -1e30f) {
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if ((go >> (4 * q)) & 1ull)
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, zz[q]), ro, so * 8,
                                                                      scratch_index(T * q, 0, M) * 8, 0);
                    }
                    pending &= ~go;
                    if (pending && !prog_spin(spins, go != 0, row_done, s - D + 1, ctrl, 1)) return;
                }
            }
            FU_T(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's ring stores are in the L2
            FU_T(4);
            if ((tid & 63) == 0) st_flag(col_done + pos * 4 + wave, s + 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) xa[q] = q < 16 - SH ? xa[q + SH] : xb[q];
        }
    } else {
        // ---------------- row workgroup of row tile `pos` ----------------
        const int fr = tid / T, rt = tid - fr * T;
        float2* __restrict__ lds = lds_all + (size_t)fr * C::SLOT;
        LdsTw<8> tw;
        lds_tw_init<8>(tw, tw256, rt);
        const int k3_0 = (int)pos * 16;
        const int xor_q = shift ? 8 : 0;
        const int e0 = scratch_index(fr, rt, M);
        for (size_t f = f_begin; f < f_end; ++f) {
            const unsigned s = (unsigned)(f - f_begin);
            FU_T(0);
            // load c comes from col workgroup c alone: it may be issued once THAT workgroup has stored frame s.
            // sc1: served by this XCD's L2, never by this CU's (possibly stale) L1
            const __amdgpu_buffer_rsrc_t ri = frame_rsrc(my_ring + (size_t)(s % D) * FU_N + (size_t)k3_0 * M, (unsigned)(16 * M * 8));
            v2f xx[16];
            {
                unsigned long long pending = 0x1111111111111111ull;
                unsigned spins = 0;
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_TRACE)
                bool first = true;
#endif
                while (pending) {
                    const unsigned long long go = NOWAIT ? pending : (ready_groups(col_done, s + 1) & pending);
#if defined(SDRK_FUSED_EXPERIMENT) && defined(FU_TRACE)
                    if (first && go) { FU_T(1); first = false; }
#endif
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if ((go >> (4 * c)) & 1ull) {
                            if ((FU_KNOCK & 4) && !(eps < -1e30f)) { xx[c] = v2f{(float)s * 1e-3f + (float)tid, (float)c}; continue; }
                            xx[c] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, e0 * 8, scratch_index(0, c * T, M) * 8, 16));
                        }
                    pending &= ~go;
                    if (pending && !prog_spin(spins, go != 0, col_done, s + 1, ctrl, 2)) return;
                }
            }
            FU_T(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's ring loads have returned
            FU_T(3);
            if ((tid & 63) == 0) st_flag(row_done + pos * 4 + wave, s + 1);
            cf v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = cf{xx[c].x, xx[c].y};
            if (!(FU_KNOCK & 8)) lds_fft_core<8, 1>(v, lds, 0, rt, tw);
            FU_T(4);
            __syncthreads();  // all rows are through their last LDS reads: the buffer becomes the transpose tile
            if (EPILOGUE == EPI_LOGPSD) {
                float* __restrict__ tile = reinterpret_cast<float*>(lds_all);  // [km][17]
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const cf z = v[rev16(q)];
                    tile[(rt + T * (q ^ xor_q)) * 17 + fr] = logpsd_db(z.x, z.y, eps);
                }
                FU_T(5);
                __syncthreads();
                const __amdgpu_buffer_rsrc_t ro = frame_rsrc(static_cast<float*>(out_raw) + f * (size_t)FU_N + k3_0,
                                                             (unsigned)((FU_N - k3_0) * 4));
                const int r = tid & 15, km0 = tid >> 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float val = tile[(km0 + T * i) * 17 + r];
                    if ((FU_KNOCK & 32) && !(eps < -1e30f || val == 12345.678f)) continue;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, (km0 * FU_A + r) * 4,
                                                          i * T * FU_A * 4, OUT_AUX);
                }
            } else {
                float2* __restrict__ tile = lds_all;  // [km][17]
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const cf z = v[rev16(q)];
                    tile[(rt + T * (q ^ xor_q)) * 17 + fr] = make_float2(z.x, z.y);
                }
                __syncthreads();
                float2* __restrict__ o = static_cast<float2*>(out_raw) + f * (size_t)FU_N + k3_0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = tid + FU_THREADS * i;
                    const int r = e & 15, km = e >> 4;
                    o[(size_t)km * FU_A + r] = tile[km * 17 + r];
                }
            }
            FU_T(6);
            __syncthreads();  // tile reads done before the next frame's exchanges
        }
    }
}

size_t fused64k_ring_bytes() { return (size_t)FU_MAX_SETS * FU_RING_SLOTS * FU_N * sizeof(float2); }
size_t fused64k_ctrl_words() { return FU_CTRL_WORDS; }
#ifdef SDRK_FUSED_EXPERIMENT
// Round-6 policy sweep (experiments/fused64k_policy/): workgroups per CU, the cache-policy bits of the streamed accesses and the
// timing-only NOWAIT build are picked per process from the environment.  Never defined in the product build.
static int fu_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
#endif
unsigned fused64k_sets(int num_cus) {
    unsigned wg_per_cu = 3;                              // three workgroups per CU are resident
#ifdef SDRK_FUSED_EXPERIMENT
    wg_per_cu = (unsigned)fu_env("SDRK_FU_WG_PER_CU", 3);
#endif
    const unsigned n = (unsigned)num_cus * wg_per_cu / 32;
    return n > (unsigned)FU_MAX_SETS ? (unsigned)FU_MAX_SETS : n;
}

// d_ctrl must hold fused64k_ctrl_words() words; it is zeroed here on the stream.
// sh = rows of the 256 x 256 view shared by consecutive frames' tiles (8: 50 % overlap, 16: none), as in launch_col.
hipError_t launch_fused64k(const LaunchArgs& a, void* d_ring, unsigned* d_ctrl) {
    if (a.n_frames == 0) return hipSuccess;
    if (a.n_frames >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const unsigned n_sets = fused64k_sets(a.num_cus), grid = n_sets * 32;
    if (n_sets == 0) return hipErrorInvalidConfiguration;
    const unsigned n_xcd = a.num_cus >= 32 ? (unsigned)a.num_cus / 32 : 1;
    const unsigned sets_per_xcd = (n_sets + n_xcd - 1) / n_xcd;
    hipError_t e = hipMemsetAsync(d_ctrl, 0, FU_CTRL_WORDS * sizeof(unsigned), a.stream);
    if (e != hipSuccess) return e;
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* twA = static_cast<const float2*>(a.d_twiddle_2p);      // W_256 (A = M = 256: twA == twM contents)
    const float2* t1T = twA + 2048 + 2048;
    const float2* t2 = t1T + (size_t)16 * FU_A;
    float2* ring = static_cast<float2*>(d_ring);
    const size_t lds_bytes = (size_t)16 * LdsCfg<8>::SLOT * sizeof(float2);   // 34,816 B (>= the [256][17] transpose tile)
    int sh = 16;
    if (a.frame_stride % (size_t)FU_A == 0 && a.frame_stride / (size_t)FU_A == 128) sh = 8;    // 50 % overlap
    if (a.frame_stride % (size_t)FU_A == 0 && a.frame_stride / (size_t)FU_A == 64) sh = 4;     // 75 % overlap
#define SDRK_FU(W, E, S)                                                                                        \
    hipLaunchKernelGGL((fused64k_kernel<W, E, S>), dim3(grid), dim3(FU_THREADS), lds_bytes, a.stream, iq,         \
                       a.frame_stride, a.d_out, (unsigned)a.n_frames, a.d_window, twA, t1T, t2, ring, d_ctrl,     \
                       n_sets, sets_per_xcd, a.eps, a.shift)
#define SDRK_FU2(W, E) do { if (sh == 8) SDRK_FU(W, E, 8); else if (sh == 4) SDRK_FU(W, E, 4); else SDRK_FU(W, E, 16); } while (0)
#ifdef SDRK_FUSED_EXPERIMENT
    if (a.epilogue == EPI_LOGPSD && a.d_window) {
        const int pin = fu_env("SDRK_FU_IN_AUX", 2), pout = fu_env("SDRK_FU_OUT_AUX", 2), nowait = fu_env("SDRK_FU_NOWAIT", 0);
        bool hit = false;
#define SDRK_FUX(I, O, NW)                                                                                      \
        if (!hit && pin == I && pout == O && nowait == (NW ? 1 : 0)) {                                            \
            hit = true;                                                                                           \
            if (sh == 8)                                                                                          \
                hipLaunchKernelGGL((fused64k_kernel<true, EPI_LOGPSD, 8, I, O, NW>), dim3(grid), dim3(FU_THREADS), \
                                   lds_bytes, a.stream, iq, a.frame_stride, a.d_out, (unsigned)a.n_frames,          \
                                   a.d_window, twA, t1T, t2, ring, d_ctrl, n_sets, sets_per_xcd, a.eps, a.shift);   \
            else                                                                                                  \
                hipLaunchKernelGGL((fused64k_kernel<true, EPI_LOGPSD, 16, I, O, NW>), dim3(grid), dim3(FU_THREADS), \
                                   lds_bytes, a.stream, iq, a.frame_stride, a.d_out, (unsigned)a.n_frames,          \
                                   a.d_window, twA, t1T, t2, ring, d_ctrl, n_sets, sets_per_xcd, a.eps, a.shift);   \
        }
#define SDRK_FUX_OUT(I, NW) SDRK_FUX(I, 2, NW) SDRK_FUX(I, 0, NW) SDRK_FUX(I, 16, NW) SDRK_FUX(I, 17, NW) SDRK_FUX(I, 19, NW)
#ifdef FU_FEW   // knock-out builds: two policies are enough
#define SDRK_FUX_IN(NW) SDRK_FUX(2, 2, NW) SDRK_FUX(2, 17, NW)
#else
#define SDRK_FUX_IN(NW) SDRK_FUX_OUT(2, NW) SDRK_FUX_OUT(0, NW) SDRK_FUX_OUT(18, NW) SDRK_FUX_OUT(19, NW)
#endif
        SDRK_FUX_IN(false)
        SDRK_FUX_IN(true)
#undef SDRK_FUX_IN
#undef SDRK_FUX_OUT
#undef SDRK_FUX
        if (!hit) return hipErrorInvalidValue;
        return hipGetLastError();
    }
#endif
    if (a.epilogue == EPI_LOGPSD) {
        if (a.d_window) SDRK_FU2(true, EPI_LOGPSD); else SDRK_FU2(false, EPI_LOGPSD);
    } else {
        if (a.d_window) SDRK_FU2(true, EPI_COMPLEX); else SDRK_FU2(false, EPI_COMPLEX);
    }
#undef SDRK_FU2
#undef SDRK_FU
    return hipGetLastError();
}

}  // namespace sdrk
