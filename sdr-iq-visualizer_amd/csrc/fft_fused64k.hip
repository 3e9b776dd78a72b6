// fft_fused64k.hip — N = 65536 (BASELINE.json config 3: the waterfall STFT) in ONE
// persistent launch whose intermediate never leaves the XCD it was produced on.
//
// Same arithmetic as a two-pass 256 x 256 split (K1 col256: DFT-256 down
// 16-wide column tiles, times W_N^(m k3); K3 row256: DFT-256 along 16 adjacent rows,
// fftshift, log epilogue) — bit-identical results — but the 512 KiB complex64
// intermediate of a frame lives in a small ring inside ONE XCD's 4 MiB L2 instead of
// crossing the fabric twice (16 of the 28 B/sample the two-launch form moves).
//
// MI355X has 8 XCDs, each with its own L2; HIP promises nothing about which XCD a
// workgroup runs on, so nothing here assumes a placement: every workgroup READS its
// XCC id (s_getreg HW_REG_XCC_ID) and joins that XCD's task queue.
//
//   queue[x]        tasks of XCD x in order: slot s = t/32, sub = t%32
//                   sub <  16 : K1 tile `sub`    of the frame in slot s     -> ring[x][s % S]
//                   sub >= 16 : K3 tile `sub-16` of the frame in slot s - 1 <- ring[x][(s-1) % S]
//   frame_of[x][s]  the workgroup that draws (s, 0) takes the next frame number from one
//                   global counter and publishes it (END once the frames run out)
//   done1/done3     per-slot completion counts of the 16 K1 / 16 K3 tiles: K3 waits for
//                   done1[s-1] == 16; K1 waits for done3[s-S] == 16 before reusing a ring slot
//
// Visibility inside one XCD: a producer's plain stores are complete in the shared L2
// once its `s_waitcnt vmcnt(0)` returns; it then bumps done1 with an agent-scope atomic.
// The consumer polls done1 (agent-scope load), then reads the ring with sc1 loads, which
// bypass its CU's L1 and are served by that same L2.  No L2 write-back is needed because
// producer and consumer were verified, by XCC id, to share the L2.
// Waits only ever point at tasks drawn earlier from the same queue, so there is no
// cycle; every spin is bounded and a timeout raises an error flag instead of hanging.
#include "fft4096_core.h"

namespace sdrk {

constexpr int FU_THREADS = 256;
constexpr int FU_N = 65536;
#ifndef FU_RING_SLOTS_N
#define FU_RING_SLOTS_N 6
#endif
#ifndef FU_LAG_N
#define FU_LAG_N 3
#endif
constexpr int FU_RING_SLOTS = FU_RING_SLOTS_N;  // x 512 KiB of each XCD's 4 MiB L2
// K3 tiles of slot s - LAG are queued with the K1 tiles of slot s: by the time a K3 tile is drawn,
// the K1 tiles it needs were drawn >= 32*LAG tasks (about one full XCD of workgroups) earlier.
constexpr int FU_LAG = FU_LAG_N;
static_assert(FU_RING_SLOTS > FU_LAG, "ring must outlast the K1 -> K3 lag");
constexpr unsigned FU_END = 0xFFFFFFFFu;
constexpr unsigned FU_POISON = 0xFFFFFFFEu;  // returned by wg_wait after a timeout / error: caller leaves
constexpr unsigned FU_SPIN_LIMIT = 1u << 18; // polls (~1-2 us each under load): give up after a fraction of a second
constexpr int FU_MAX_XCD = 16;

// control block layout (unsigned words), zeroed before every launch
//   [0] next frame   [1] error flag   [32 + x*XS ...] per XCD: [0] queue head, then at +32:
//   frame_of[max_slots], done1[max_slots], done3[max_slots]
__host__ __device__ inline size_t fu_xcd_stride(unsigned max_slots) { return 32 + 3 * (size_t)((max_slots + 31) & ~31u); }
__host__ inline size_t fused64k_ctrl_words(unsigned max_slots) { return 32 + FU_MAX_XCD * fu_xcd_stride(max_slots); }

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One parallel poll of up to three control words by lanes 0..2 of wave 0 (a single wave
// instruction, one round trip); lane i spins until its word satisfies its condition
// (want == 0: non-zero; else >= want).  Results are broadcast through LDS; FU_POISON in
// any slot means a timeout or a raised error flag and the caller leaves.
struct Poll3 {
    unsigned v[3];
};
__device__ __forceinline__ Poll3 wg_poll3(const unsigned* p0, unsigned w0, const unsigned* p1, unsigned w1,
                                          const unsigned* p2, unsigned w2, unsigned* err, unsigned* sh) {
    const int tid = threadIdx.x;
    if (tid < 3) {
        const unsigned* p = tid == 0 ? p0 : (tid == 1 ? p1 : p2);
        const unsigned want = tid == 0 ? w0 : (tid == 1 ? w1 : w2);
        unsigned v = p ? ld_agent(p) : 1u, spins = 0;
        while (p && want == 0 && v == 0) {   // only the publish words spin here; a count is sampled once
            __builtin_amdgcn_s_sleep(4);
            if (++spins > FU_SPIN_LIMIT || ld_agent(err)) {
                if (!ld_agent(err)) {  // first reporter leaves a record: which word, what it held
                    err[1] = (unsigned)(p - err); err[2] = v; err[3] = want; err[4] = 100u + tid;
                }
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = FU_POISON;
                break;
            }
            v = ld_agent(p);
        }
#ifdef FU_STATS
        if (spins) atomicAdd(err + 8 + tid, spins);      // record[8..10]: spins on cur / prev chunk words
#endif
        sh[tid] = v;
    }
    __syncthreads();
    Poll3 r;
    r.v[0] = sh[0]; r.v[1] = sh[1]; r.v[2] = sh[2];
    __syncthreads();
    return r;
}

// Thread 0 spins until *p >= want; workgroup-uniform result (FU_POISON on timeout / error).
__device__ __forceinline__ unsigned wg_wait_count(const unsigned* p, unsigned want, unsigned* err, unsigned* sh,
                                                  int site = 1) {
    if (threadIdx.x == 0) {
        unsigned v = ld_agent(p), spins = 0;
        while (v < want) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > FU_SPIN_LIMIT || ld_agent(err)) {
                if (!ld_agent(err)) {
                    err[1] = (unsigned)(p - err); err[2] = v; err[3] = want; err[4] = 200u;
                }
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = FU_POISON;
                break;
            }
            v = ld_agent(p);
        }
#ifdef FU_STATS
        if (spins) atomicAdd(err + 11 + site, spins);   // record[12]: K1 ring-slot wait spins, record[13]: K3 done1 wait spins
        atomicAdd(err + 13 + site, 1u);      // record[14], [15]: how many such waits
#endif
        sh[0] = v;
    }
    __syncthreads();
    const unsigned v = sh[0];
    __syncthreads();
    return v;
}

template <bool HAS_WINDOW, int EPILOGUE>
__global__ __launch_bounds__(FU_THREADS, 3) void fused64k_kernel(
    const float2* __restrict__ iq, size_t frame_stride, void* __restrict__ out_raw, unsigned n_frames,
    const float* __restrict__ window, const float2* __restrict__ tw4096, const float2* __restrict__ t1,
    const float2* __restrict__ t2, float2* __restrict__ ring, unsigned* __restrict__ ctrl, unsigned max_slots,
    float eps, int shift) {
    __shared__ float2 lds[16 * 272 + 256];
    __shared__ unsigned sh_u[4];
    float2* __restrict__ tw256 = lds + 16 * 272;
    const int tid = threadIdx.x;
    const int lo = tid & 15, hi = tid >> 4;
    tw256[tid] = tw4096[(16 * lo * hi) & 4095];

    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= FU_MAX_XCD - 1;
    unsigned* __restrict__ err = ctrl + 1;
    unsigned* __restrict__ xc = ctrl + 32 + xcc * fu_xcd_stride(max_slots);
    unsigned* __restrict__ queue = xc;
    const size_t arr = (max_slots + 31) & ~31u;
    unsigned* __restrict__ frame_of = xc + 32;
    unsigned* __restrict__ done1 = frame_of + arr;
    unsigned* __restrict__ done3 = done1 + arr;
    float2* __restrict__ my_ring = ring + (size_t)xcc * FU_RING_SLOTS * FU_N;
    __syncthreads();

    // exchange-1 style addressing for K1 (XOR swizzle), padded layout for K3
    const int b = hi & 1;
    const int x1r_even = lo + 256 * hi + 16 * b, x1r_odd = lo + 256 * hi - 16 * b;
    const int xor_q = shift ? 8 : 0;

    // Frames are claimed 16 at a time: the workgroup that draws (s, 0) with s % 16 == 0 takes the
    // next 16 frame numbers from the global counter and publishes base+1 in chunk_base[s / 16].
    unsigned* __restrict__ chunk_base = frame_of;  // indexed by s >> 4 (array is max_slots long: ample)

    // A workgroup can never legitimately draw more tasks than one XCD's queue holds.
    const unsigned max_draws = max_slots * 32u;
    unsigned t_next = 0;
    if (tid == 0) t_next = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned draws = 0;; ++draws) {
        if (tid == 0) sh_u[3] = t_next;
        __syncthreads();
        const unsigned t = sh_u[3];
        __syncthreads();
        if (draws > max_draws) {
            if (tid == 0) __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const unsigned s = t >> 5, sub = t & 31;
        if (s >= max_slots) {  // cannot happen with the host's sizing; never index past the arrays
            if (tid == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (sub == 0 && (s & 15) == 0 && tid == 0) {
            const unsigned f0 = __hip_atomic_fetch_add(ctrl, 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(chunk_base + (s >> 4), f0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the next task id is fetched while this one runs (its latency hides under the data loads)
        if (tid == 0) t_next = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // one round trip: both chunk words and the completion count this task depends on
        const unsigned* dep = nullptr;
        if (sub < 16) { if (s >= FU_RING_SLOTS) dep = done3 + s - FU_RING_SLOTS; }
        else if (s >= FU_LAG) dep = done1 + s - FU_LAG;
        const Poll3 pl = wg_poll3(chunk_base + (s >> 4), 0, s >= FU_LAG ? chunk_base + ((s - FU_LAG) >> 4) : nullptr, 0,
                                  dep, 16, err, sh_u);
        if (pl.v[0] == FU_POISON || pl.v[1] == FU_POISON) break;
        const bool dep_ready = dep == nullptr || pl.v[2] >= 16;   // sampled once alongside; waited for below if needed
        const unsigned fc = pl.v[0] - 1 + (s & 15);
        const unsigned fp = s >= FU_LAG ? pl.v[1] - 1 + ((s - FU_LAG) & 15) : 0;
        const unsigned cur = fc < n_frames ? fc + 1 : FU_END;
        const unsigned prev = (s >= FU_LAG && fp < n_frames) ? fp + 1 : FU_END;
        // Finished only when the frames ran out at least LAG slots ago: then every later task is void too
        // (K1 of an END slot; K3 of a slot >= the first END slot).  For s < LAG there is no "prev" slot yet.
        if (cur == FU_END && s >= FU_LAG && prev == FU_END) {
            // The task already drawn ahead is void too (frames only run out once), but if it carries
            // the duty to publish a chunk word, tasks queued behind it are waiting for that word.
            if (tid == 0) {
                const unsigned s2 = t_next >> 5;
                if ((t_next & 31) == 0 && (s2 & 15) == 0 && s2 < max_slots) {
                    const unsigned f0 = __hip_atomic_fetch_add(ctrl, 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(chunk_base + (s2 >> 4), f0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            break;
        }

        if (sub < 16) {
            // ---------------- K1: column tile `sub` of frame cur-1 -> ring slot s % S ----------------
            if (cur == FU_END) continue;
            const size_t f = cur - 1;
            const int m = (int)sub * 16 + lo;
            const float2* __restrict__ x = iq + f * frame_stride + m;
            cf v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int off = (hi + 16 * j) * 256;
                v2f tt = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(x + off));
                if (HAS_WINDOW) {
                    const float w = window[off + m];
                    tt.x *= w;
                    tt.y *= w;
                }
                v[j] = cf{tt.x, tt.y};
            }
            radix16(v);
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
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                float2 tt = lds[((a & 1) ? x1r_odd : x1r_even) + 16 * a];
                v[a] = cf{tt.x, tt.y};
            }
            radix16(v);
            // the ring slot must have been drained by the K3 tiles of slot s - S (usually long ago)
            if (!dep_ready && wg_wait_count(dep, 16, err, sh_u, 1) == FU_POISON) break;
            float2* __restrict__ o = my_ring + (size_t)(s % FU_RING_SLOTS) * FU_N + m;
            const float2 bw = t1[m * 16 + hi];
            const cf base = cf{bw.x, bw.y};
            const float4* __restrict__ row = reinterpret_cast<const float4*>(t2 + (size_t)m * 16);
#pragma unroll
            for (int q2 = 0; q2 < 8; ++q2) {
                const float4 w = row[q2];
                cf z0 = cmul(v[rev16(2 * q2)], cmul(base, cf{w.x, w.y}));
                cf z1 = cmul(v[rev16(2 * q2 + 1)], cmul(base, cf{w.z, w.w}));
                o[(hi + 32 * q2) * 256] = make_float2(z0.x, z0.y);
                o[(hi + 32 * q2 + 16) * 256] = make_float2(z1.x, z1.y);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's ring stores are in the L2
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(done1 + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // ---------------- K3: row tile `sub-16` of frame prev-1 <- ring slot (s-1) % S ----------------
            if (prev == FU_END) continue;
            if (!dep_ready && wg_wait_count(dep, 16, err, sh_u, 2) == FU_POISON) break;
            const size_t f = prev - 1;
            const int k3_0 = (int)(sub - 16) * 16;
            const unsigned long long* __restrict__ in = reinterpret_cast<const unsigned long long*>(
                my_ring + (size_t)((s - FU_LAG) % FU_RING_SLOTS) * FU_N + (size_t)(k3_0 + hi) * 256 + lo);
            cf v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // sc1: served by this XCD's L2, never by this CU's (possibly stale) L1
                const unsigned long long raw = __hip_atomic_load(in + 16 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const v2f tt = __builtin_bit_cast(v2f, raw);
                v[j] = cf{tt.x, tt.y};
            }
            radix16(v);
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
            // all of this workgroup's ring reads have returned: the slot may be reused once 16 tiles say so
            if (tid == 0) __hip_atomic_fetch_add(done3 + s - FU_LAG, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float2 tt = lds[u + 17 * lo + 272 * hi];
                v[u] = cf{tt.x, tt.y};
            }
            radix16(v);
            if (EPILOGUE == EPI_LOGPSD) {
                float* __restrict__ o = static_cast<float*>(out_raw) + f * (size_t)FU_N + k3_0 + lo;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k1 = hi + 16 * (q ^ xor_q);
                    cf z = v[rev16(q)];
                    __builtin_nontemporal_store(logpsd_db(z.x, z.y, eps), &o[256 * k1]);
                }
            } else {
                float2* __restrict__ o = static_cast<float2*>(out_raw) + f * (size_t)FU_N + k3_0 + lo;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k1 = hi + 16 * (q ^ xor_q);
                    cf z = v[rev16(q)];
                    o[256 * k1] = make_float2(z.x, z.y);
                }
            }
            __syncthreads();  // LDS is reused by the next task
        }
    }
}

size_t fused64k_ring_bytes() { return (size_t)FU_MAX_XCD * FU_RING_SLOTS * FU_N * sizeof(float2); }

unsigned fused64k_max_slots(size_t n_frames, unsigned grid) { return (unsigned)(n_frames + grid / 16 + 8 + FU_LAG); }

size_t fused64k_ctrl_words_for(size_t n_frames, int num_cus) {
    return fused64k_ctrl_words(fused64k_max_slots(n_frames, (unsigned)num_cus * 3));
}

// d_ctrl must hold fused64k_ctrl_words(max_slots) words; it is zeroed here on the stream.
hipError_t launch_fused64k(const LaunchArgs& a, void* d_ring, unsigned* d_ctrl, size_t ctrl_capacity_words) {
    if (a.n_frames == 0) return hipSuccess;
    if (a.n_frames >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)a.num_cus * 3;
    const unsigned max_slots = fused64k_max_slots(a.n_frames, grid);
    const size_t words = fused64k_ctrl_words(max_slots);
    if (words > ctrl_capacity_words) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_ctrl, 0, words * sizeof(unsigned), a.stream);
    if (e != hipSuccess) return e;
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
    const float2* t1 = static_cast<const float2*>(a.d_twiddle_fused);
    const float2* t2 = t1 + 256 * 16;
    float2* ring = static_cast<float2*>(d_ring);
#define SDRK_FU(W, E)                                                                                  \
    hipLaunchKernelGGL((fused64k_kernel<W, E>), dim3(grid), dim3(FU_THREADS), 0, a.stream, iq,          \
                       a.frame_stride, a.d_out, (unsigned)a.n_frames, a.d_window, tw, t1, t2, ring, d_ctrl, \
                       max_slots, a.eps, a.shift)
    if (a.epilogue == EPI_LOGPSD) {
        if (a.d_window) SDRK_FU(true, EPI_LOGPSD); else SDRK_FU(false, EPI_LOGPSD);
    } else {
        if (a.d_window) SDRK_FU(true, EPI_COMPLEX); else SDRK_FU(false, EPI_COMPLEX);
    }
#undef SDRK_FU
    return hipGetLastError();
}

}  // namespace sdrk
