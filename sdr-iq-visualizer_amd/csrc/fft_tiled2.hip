// fft_tiled2.hip — frames of N = A * M samples, 2^15 <= N <= 2^22, in TWO tiled passes whose
// sub-transforms (A, M <= 2048 points) run entirely in registers + LDS (fft_lds_core.h):
// window -> FFT -> fftshift -> 20*log10(|X|+eps)   (app/sdr/streamer.py:119,121); the
// waterfall sizes of BASELINE.json configs 3 (N = 65536) and 5 (N = 2^20).
//
//   n = m + M n3 (m < M contiguous, n3 < A),   k = A km + k3
//   X[A km + k3] = sum_m W_M^(m km) * W_N^(m k3) * [ sum_n3 x[m + M n3] W_A^(n3 k3) ]
//
//   col pass   tile = 16 adjacent m x all A values of n3 (stride M): DFT-A per column, sixteen
//              columns interleaved in LDS (lanes run along m: 128-byte segments in and out),
//              times W_N^(m k3) = t1T[tau][m] * t2[m][q]  (k3 = tau + (A/16) q)   -> scratch (k3, m), stored
//              in 16 x 16 squares (scratch_index below)
//   row pass   tile = 16 adjacent k3 rows x M (each row contiguous): DFT-M per row, fftshift
//              (km + M/2), log epilogue, then a 16 x M transpose through LDS so that the stores
//              X[A km + k3] run along k3 (64-byte segments); M >= 1024: the 16 rows as two 8-row halves
//              (row_pass_pair_kernel)
// 28 B/sample of traffic (8 in, 8+8 scratch, 4 out); the scratch is processed in chunks that stay in the
// 256 MiB Infinity Cache between the pass that writes them and the pass that reads them.
#include "fft_lds_core.h"

// cache-policy bits of the four streams (buffer `aux` operand on gfx950: 1 = sc0, 2 = nt, 16 = sc1).  The defaults are
// what ships; the macros exist for A/B builds (tools/variant.sh) — round 4 measured every combination that names a
// different path through the L2 / Infinity Cache for the scratch.
#ifndef SDRK_COL_LD_AUX
#define SDRK_COL_LD_AUX 2    // input samples: read once
#endif
#ifndef SDRK_SCR_ST_AUX
#define SDRK_SCR_ST_AUX 0    // scratch written by the col pass ...
#endif
#ifndef SDRK_SCR_LD_AUX
#define SDRK_SCR_LD_AUX 0    // ... and read back by the row pass of the same chunk
#endif
#ifndef SDRK_ROW_ST_AUX
#define SDRK_ROW_ST_AUX 2    // dB rows: written once
#endif
// a row of M <= 1024 points is transformed by T = M / 16 <= 64 threads, i.e. inside one wave: its exchanges need no
// workgroup barrier (SDRK_ROW_WAVE_SYNC=0 in A/B builds restores them)
#ifndef SDRK_ROW_WAVE_SYNC
#define SDRK_ROW_WAVE_SYNC 1
#endif
#define SDRK_ROW_SYNC(T) ((SDRK_ROW_WAVE_SYNC && (T) <= 64) ? 2 : 0)

namespace sdrk {

// (scratch_index, the layout of the intermediate between the passes, lives in fft_lds_core.h)

// W = tile width in columns (16, or 8 for A = 2048 so that the tile fits the LDS).
// FIXED: the grid is a multiple of the tiles per frame, so every workgroup keeps the same tile position (same
// m) for all its frames.  Its window coefficients and its W_N^(m k3) factors are then loop invariants held in
// registers, and the kernel runs a software pipeline: the next tile's 16 loads per thread are issued before the
// current tile's passes and stay in flight across its barriers, so that the load latency, the transform and the
// scratch stores of consecutive tiles overlap inside ONE workgroup instead of relying on the co-resident
// workgroups being in different phases (measured on config 3: the col pass alone takes 2.3 ms without its
// stores, 2.4 ms without its loads and 3.8 ms with both — serialised phases).  Nothing in the loop body other
// than the stream itself touches vector memory: vmcnt counts loads and stores in issue order on gfx950, so any
// table load issued after the stores would make the wave wait for the store acknowledgements.
//
// SH (FIXED only): overlapped frames cut from one stream (the STFT of config 3) share samples, and when the hop
// is SH * T rows of the A x M view (SH = 8: 50 % overlap, SH = 4: 75 %), row n3 of frame f+1 is row n3 + SH*T of
// frame f — the SAME thread's load number q + SH.  Each workgroup then walks a run of consecutive frames of its
// tile position and loads only the SH new values per thread and frame, shifting the other 16 - SH down in
// registers: the col pass reads every input sample once instead of 16/SH times (PMC, 384-frame chunks at 50 %
// overlap: 150.8 MB fetched per launch before — L2 caught half of the re-reads — against 100.9 MB unique).
// SH = 16: no reuse (packed frames or any other hop).
template <int LOG2A, bool HAS_WINDOW, int W, bool FIXED, int SH>
__global__ __launch_bounds__(LdsCfg<LOG2A>::T * W, (LdsCfg<LOG2A>::T * W >= 512 ? 4 : 3)) void col_pass_kernel(
    const float2* __restrict__ iq, size_t frame_stride, float2* __restrict__ scratch, size_t n_frames, int M,
    const float* __restrict__ window, const float2* __restrict__ twA, const float2* __restrict__ t1T,
    const float2* __restrict__ t2, unsigned in_bytes) {   // in_bytes: bytes of each input frame that exist (the rest reads as zeros)
    using C = LdsCfg<LOG2A>;
    constexpr int A = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0;
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];  // 17/16 A W elements: W interleaved columns
    const int tid = threadIdx.x;
    const int fr = tid & (W - 1), tau = tid / W;
    LdsTw<LOG2A> tw;
    lds_tw_init<LOG2A>(tw, twA, tau);
    const size_t nfft = (size_t)A * M;
    const int tiles = M / W;
    const size_t items = n_frames * (size_t)tiles;
    const int estep = T * M;             // n3 += T

    // tile (frame f, columns m .. ) of work item g
    auto locate = [&](size_t g, size_t& f, int& m) {
        // W == 8: the two tiles that share each 128-byte line go to blocks b and b+8 (same XCD under the
        // round-robin placement; a speed hint only)
        size_t it = g;
        if (W == 8 && (items & 15) == 0) it = (g & ~(size_t)15) + ((g & 7) << 1) + ((g >> 3) & 1);
        f = it / tiles;
        m = (int)(it - f * tiles) * W + fr;
    };
    // B[k3 = tau + T q] * W_N^(m k3),  W_N^(m k3) = W_N^(m tau) * W_N^(m T q) = t1T[tau][m] * t2[m][q]
    auto factors = [&](int m, cf (&bw)[16]) {
        const float2 b0 = t1T[tau * M + m];
        const cf base = cf{b0.x, b0.y};
        const float4* __restrict__ row = reinterpret_cast<const float4*>(t2 + (size_t)m * 16);
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) {
            const float4 w = row[q2];
            bw[2 * q2] = cmul(base, cf{w.x, w.y});
            bw[2 * q2 + 1] = cmul(base, cf{w.z, w.w});
        }
    };
    // the 16 loads of one tile (raw samples).  Buffer addressing: wave-uniform descriptor on the frame (zero-sized
    // when there is no such item: the loads return zeros without touching memory), one 32-bit lane offset,
    // uniform row steps
    auto issue = [&](size_t g, v2f (&x)[16]) {
        size_t f = 0;
        int m = 0;
        const bool live = g < items;
        if (live) locate(g, f, m);
        const __amdgpu_buffer_rsrc_t rx = frame_rsrc(iq + f * frame_stride, live ? in_bytes : 0u);
        const int e0 = tau * M + m;          // element (n3 = tau, m)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            x[q] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, e0 * 8, q * estep * 8, SDRK_COL_LD_AUX));
    };
    auto store = [&](size_t f, int m, const cf (&v)[16], const cf (&bw)[16]) {
        const __amdgpu_buffer_rsrc_t ro = frame_rsrc(scratch + f * nfft, (unsigned)(nfft * 8));
        // k3 = tau + T q: lane part of the scratch index from (tau, m), uniform part from T q (both layouts are
        // additive in that split because T q is a multiple of 16 or, for T = 8, tau + 8 stays inside one square)
        const int so = scratch_index(tau, m, M);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const cf z = cmul(v[rev16(q)], bw[q]);
            const v2f sv = {z.x, z.y};
            const int u = scratch_index(T * q, 0, M);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, sv), ro, so * 8, u * 8, SDRK_SCR_ST_AUX);
        }
    };

    if (FIXED) {
        // tile position t = blockIdx % tiles; the gridDim / tiles workgroups of one position split the frames
        // into contiguous runs
        // W == 8: positions p and p + 8 (workgroups b and b + 8: the same XCD under the round-robin placement) share
        // the 128-byte lines of input and scratch
        const int p = (int)(blockIdx.x % tiles);
        const int m = ((W == 8 && (tiles & 15) == 0) ? (p & ~15) + ((p & 7) << 1) + ((p >> 3) & 1) : p) * W + fr;
        const size_t lanes = gridDim.x / tiles, lane = blockIdx.x / tiles;
        const size_t run = (n_frames + lanes - 1) / lanes;
        const size_t f_begin = lane * run, f_end = f_begin + run < n_frames ? f_begin + run : n_frames;
        if (f_begin >= f_end) return;
        float wreg[16];
        if (HAS_WINDOW) {
#pragma unroll
            for (int q = 0; q < 16; ++q) wreg[q] = window[(size_t)(tau + T * q) * M + m];
        }
        cf bw[16];
        factors(m, bw);
        const int e0 = tau * M + m;          // element (n3 = tau, m)
        // loads q >= q0 of frame f (zero-sized descriptor past the run: zeros, no memory access)
        auto issue_from = [&](size_t f, v2f (&x)[16], int q0) {
            const bool live = f < f_end;
            const __amdgpu_buffer_rsrc_t rx = frame_rsrc(iq + (live ? f : f_begin) * frame_stride, live ? in_bytes : 0u);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q >= q0) x[q] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, e0 * 8, q * estep * 8, SDRK_COL_LD_AUX));
        };
        v2f xa[16], xb[16];
        issue_from(f_begin, xa, 0);
        for (size_t f = f_begin; f < f_end; ++f) {
            issue_from(f + 1, xb, 16 - SH);
            cf v[16];
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const int q = i + C0 * j;
                    v[i * R0 + j] = HAS_WINDOW ? cf{xa[q].x, xa[q].y} * wreg[q] : cf{xa[q].x, xa[q].y};
                }
            lds_fft_core<LOG2A, W>(v, lds_all, fr, tau, tw);
            store(f, m, v, bw);
#pragma unroll
            for (int q = 0; q < 16; ++q) xa[q] = q < 16 - SH ? xa[q + SH] : xb[q];
        }
    } else {
        const __amdgpu_buffer_rsrc_t rw = frame_rsrc(window, HAS_WINDOW ? (unsigned)(nfft * 4) : 0u);
        for (size_t g = blockIdx.x; g < items; g += gridDim.x) {
            size_t f;
            int m;
            locate(g, f, m);
            v2f x[16];
            issue(g, x);
            const int e0 = tau * M + m;
            cf v[16];
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const int q = i + C0 * j;
                    v2f t = x[q];
                    if (HAS_WINDOW) {
                        const float w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, e0 * 4, q * estep * 4, 0));
                        t.x *= w;
                        t.y *= w;
                    }
                    v[i * R0 + j] = cf{t.x, t.y};
                }
            lds_fft_core<LOG2A, W>(v, lds_all, fr, tau, tw);
            cf bw[16];
            factors(m, bw);
            store(f, m, v, bw);
        }
    }
}

// col pass with a staged fetch, for the sizes whose workgroups are too large for the register pipeline above:
// A = 512 (N = 2^18, 2^19; W = 16 columns) and A = 1024 (N = 2^20, 2^21; W = 8).  A sixteen-column tile of A = 1024
// is 128 KiB and leaves neither LDS nor registers for a second tile, so fetch, transform and store of the generic
// kernel run one after the other; at A = 512 two such workgroups per CU overlap only by chance.  This kernel takes
// 64 KiB of samples per tile and spends the other half of the LDS on a staging buffer for the NEXT tile's raw
// samples, filled by LDS-DMA while the current tile is being transformed:
//   LDS = [exchange area 17/16 x A x W complex | staging A rows x 8 W bytes]                  (133.5 KiB, 512 threads)
//   per tile:  wait for the staged samples -> pick them up (x window) -> start the DMA of the next tile ->
//              DFT-A in registers + exchange area (raw barriers: a __syncthreads() would drain the DMA) ->
//              x W_N^(m k3) -> 16 stores per thread
// A workgroup keeps one tile position for a run of consecutive frames (grid = a multiple of the M/W positions on a
// full device), so its window coefficients and factors are registers for the whole run and the loop issues no
// vector-memory instruction besides the DMA and the stores; the wait at the top of a tile is therefore an exact
// vmcnt(16): the DMA is older than the previous tile's 16 stores, which may still be in flight.
// The DMA is issued from inline assembly.  The compiler orders every LDS access that follows an LDS-DMA it knows
// about behind a vmcnt(0) — also the accesses to the exchange area, which the DMA never writes — and that wait put
// the whole fetch (and the previous tile's stores) back in front of the transform: at N = 2^20, 105.8 us per
// 24-frame chunk with the builtin against 96.3 us like this (generic kernel: 107).  Timing-only builds there:
// transform alone 34 us, + stores 49, + fetch 67, all three 96; FETCH_SIZE / WRITE_SIZE = 211 / 196 MB per chunk,
// i.e. the algorithmic bytes: for W = 8 the two tiles that share each 128-byte line of input and of scratch go to
// workgroups b and b + 8, the same XCD under the observed round-robin placement, so the second half of a line is
// an L2 hit (a speed hint only; nothing depends on it).  With whole lines (A = 512, W = 16) the same structure
// reaches 81.8 us per chunk of the same size at N = 2^18 (generic kernel: 94.0), 5.0 TB/s of streamed bytes.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // M0 is named as clobbered on purpose
template <int LOG2A, bool HAS_WINDOW, int W>
__global__ __launch_bounds__(LdsCfg<LOG2A>::T * W, 2) void col_pass_staged_kernel(
    const float2* __restrict__ iq, size_t frame_stride, float2* __restrict__ scratch, size_t n_frames, int M,
    const float* __restrict__ window, const float2* __restrict__ twA, const float2* __restrict__ t1T,
    const float2* __restrict__ t2, unsigned in_bytes) {
    using C = LdsCfg<LOG2A>;
    constexpr int A = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0;
    constexpr int LPR = W / 2;                                            // lanes per row: 16 bytes each
    constexpr int WAVES = T * W / 64, ROWS_PER_INSTR = 64 / LPR;
    constexpr int INSTR = A / (ROWS_PER_INSTR * WAVES);                  // DMA instructions per wave and tile
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];
    float2* __restrict__ xch = lds_all;                                   // SLOT * W elements
    float2* __restrict__ stage = lds_all + C::SLOT * W;                   // A * W elements, row n3 at n3 * W
    const int tid = threadIdx.x;
    const int fr = tid & (W - 1), tau = tid / W;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    LdsTw<LOG2A> tw;
    lds_tw_init<LOG2A>(tw, twA, tau);
    const size_t nfft = (size_t)A * M;
    const int tiles = M / W;
    // work units: (tile position, run of frames); one per workgroup when the grid is a multiple of `tiles`
    const size_t lanes = gridDim.x >= (unsigned)tiles ? gridDim.x / tiles : 1;
    const size_t run = (n_frames + lanes - 1) / lanes;
    typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
    for (size_t unit = blockIdx.x; unit < lanes * tiles; unit += gridDim.x) {
        const int p = (int)(unit % tiles);
        // W == 8: positions b and b + 8 share the 128-byte lines
        const int t = (W == 8 && (tiles & 15) == 0) ? (p & ~15) + ((p & 7) << 1) + ((p >> 3) & 1) : p;
        const int m0 = t * W, m = m0 + fr;
        const size_t ln = unit / tiles;
        const size_t f_begin = ln * run, f_end = f_begin + run < n_frames ? f_begin + run : n_frames;
        if (f_begin >= f_end) continue;

        float wreg[16];
        if (HAS_WINDOW) {
#pragma unroll
            for (int q = 0; q < 16; ++q) wreg[q] = window[(size_t)(tau + T * q) * M + m];
        }
        cf bw[16];   // W_N^(m k3) for k3 = tau + T q
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
        const int voff = ((lane / LPR) * M + m0) * 8 + (lane % LPR) * 16;   // row lane/LPR of the group, 16-byte piece lane%LPR
        auto dma = [&](size_t f) {
            if (f >= f_end) return;                  // nothing may still be writing LDS when the workgroup exits
            const unsigned long long base = (unsigned long long)(iq + f * frame_stride);
            const v4u32 rx = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, in_bytes, 0x00020000u};
#pragma unroll
            for (int i = 0; i < INSTR; ++i) {
                const int grp = wave + WAVES * i;                          // rows ROWS_PER_INSTR * grp ...
                const unsigned lds_addr =
                    (unsigned)(size_t)(__attribute__((address_space(3))) void*)(stage + grp * (ROWS_PER_INSTR * W));
                // M0 = LDS base of the wave's 1 KiB; one wait state between the M0 write and the LDS-DMA
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                             :: "s"(lds_addr), "v"(voff), "s"(rx), "s"(grp * ROWS_PER_INSTR * M * 8) : "memory", "m0");
            }
        };
        if (unit != blockIdx.x) {                    // (partial devices only: more than one unit per workgroup)
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the table loads above must not be counted as stores
            lds_core_barrier<true>();
        }
        dma(f_begin);
        const int so = scratch_index(tau, m, M);
        for (size_t f = f_begin; f < f_end; ++f) {
            // the staged tile has landed: everything older than the previous tile's 16 stores is complete
            if (f == f_begin) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x4F70);                       // vmcnt(16)
            lds_core_barrier<true>();                                      // ... in every wave
            cf v[16];
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const int q = i + C0 * j;
                    const float2 x = stage[(tau + T * q) * W + fr];
                    v[i * R0 + j] = HAS_WINDOW ? cf{x.x, x.y} * wreg[q] : cf{x.x, x.y};
                }
            lds_core_barrier<true>();                                      // every wave has picked its samples up
            dma(f + 1);
            lds_fft_core<LOG2A, W, 1>(v, xch, fr, tau, tw);
            const __amdgpu_buffer_rsrc_t ro = frame_rsrc(scratch + f * nfft, (unsigned)(nfft * 8));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = cmul(v[rev16(q)], bw[q]);
                const v2f sv = {z.x, z.y};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, sv), ro, so * 8, scratch_index(T * q, 0, M) * 8, SDRK_SCR_ST_AUX);
            }
        }
    }
}
#pragma clang diagnostic pop

#ifndef SDRK_COL_W256
#define SDRK_COL_W256 16
#endif
// A = 512 (N = 2^18 ... 2^20): SDRK_STAGED_W9 columns through col_pass_staged_kernel (16: one 512-thread workgroup per
// CU; 8: two 256-thread workgroups per CU), or, with SDRK_STAGED_W9 = 0, SDRK_COL_W9 columns through col_pass_kernel
// (8: software-pipelined 256-thread workgroups, three per CU)
#ifndef SDRK_STAGED_W9
#define SDRK_STAGED_W9 16
#endif
#ifndef SDRK_COL_W9
#define SDRK_COL_W9 16
#endif
// columns per col-pass tile; A = 512 and A = 1024 take col_pass_staged_kernel with STAGED_W columns
#define STAGED_W(LOG2A) ((LOG2A) == 10 ? 8 : ((LOG2A) == 9 ? SDRK_STAGED_W9 : 0))
#define COL_TILE_W(LOG2A) (STAGED_W(LOG2A) ? STAGED_W(LOG2A) : ((LOG2A) == 11 ? 8 : ((LOG2A) == 8 ? SDRK_COL_W256 : ((LOG2A) == 9 ? SDRK_COL_W9 : 16))))
// the software-pipelined (FIXED) form of col_pass_kernel: workgroups of <= 256 threads, whole lines (16 columns) or paired half lines
__host__ __device__ constexpr bool col_can_fix(int T, int W) { return T * W <= 256 && (W >= 16 || (W == 8 && T == 32)); }
// workgroups of col_pass_staged_kernel per CU: LDS = exchange area + staging buffer
__host__ __device__ constexpr int staged_per_cu(int SLOT, int A, int W) {
    return (size_t)2 * ((size_t)SLOT * W + (size_t)A * W) * 8 <= 160 * 1024 ? 2 : 1;
}

// row pass for M = 256 and 512: tile = 16 adjacent k3 rows x M (34.8 / 69.6 KiB of LDS: four / two workgroups per
// CU, which overlap each other's load, transform and store phases).
// The complex epilogues of the row passes: plain (EPI_COMPLEX) or with the chirp-z multiplies riding on the stores.
//   k = natural output index A km + k3 of the value (the EPI_BLU_* forms run unshifted), f = frame, nfft = A M.
struct EpiTab {
    const float2* tab;   // EPI_BLU_MUL: filter spectrum [nfft]; EPI_BLU_POST_*: chirp [n_out]
    int n_out;           // EPI_BLU_POST_*: values per output row
    int rot;             //                 fftshift rotation (n_out / 2 or 0)
    float inv_m;         //                 1 / nfft
};
template <int EPILOGUE>
__device__ __forceinline__ void epi_store_complex(void* __restrict__ out_raw, size_t f, size_t nfft, size_t k, float2 v,
                                                  const EpiTab& t, float eps) {
    if (EPILOGUE == EPI_COMPLEX) {
        static_cast<float2*>(out_raw)[f * nfft + k] = v;
    } else if (EPILOGUE == EPI_BLU_MUL) {
        const float2 b = t.tab[k];
        static_cast<float2*>(out_raw)[f * nfft + k] = make_float2(fmaf(v.x, b.x, -(v.y * b.y)), -fmaf(v.x, b.y, v.y * b.x));  // conj(v b)
    } else {
        if (k >= (size_t)t.n_out) return;
        const float2 c = t.tab[k];
        const float re = fmaf(c.x, v.x, -(c.y * v.y)) * t.inv_m;                   // conj(c v) / M, as blu_post_kernel forms it
        const float im = -fmaf(c.x, v.y, c.y * v.x) * t.inv_m;
        size_t dst = k + (size_t)t.rot;
        if (dst >= (size_t)t.n_out) dst -= (size_t)t.n_out;
        if (EPILOGUE == EPI_BLU_POST_LOG)
            static_cast<float*>(out_raw)[f * (size_t)t.n_out + dst] = logpsd_db(re, im, eps);
        else
            static_cast<float2*>(out_raw)[f * (size_t)t.n_out + dst] = make_float2(re, im);
    }
}

template <int LOG2M, int EPILOGUE>
__global__ __launch_bounds__(LdsCfg<LOG2M>::T * 16, (LdsCfg<LOG2M>::T * 16 >= 512 ? 4 : 3)) void row_pass_kernel(
    const float2* __restrict__ scratch, void* __restrict__ out_raw, size_t n_frames, int A,
    const float2* __restrict__ twM, float eps, int shift, EpiTab epi) {
    using C = LdsCfg<LOG2M>;
    constexpr int M = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0, ROWS = 16;
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];  // 16 rows x SLOT, reused for the transpose
    const int tid = threadIdx.x;
    const int fr = tid / T, rt = tid - fr * T;
    float2* __restrict__ lds = lds_all + (size_t)fr * C::SLOT;
    LdsTw<LOG2M> tw;
    lds_tw_init<LOG2M>(tw, twM, rt);
    const size_t nfft = (size_t)A * M;
    const int tiles = A / ROWS;
    const size_t items = n_frames * (size_t)tiles;
    const int xor_q = shift ? 8 : 0;
    const int e0 = scratch_index(fr, rt, M);          // element (k3_0 + fr, m = rt + T c): lane part; uniform part from T c
    for (size_t it = blockIdx.x; it < items; it += gridDim.x) {
        const size_t f = it / tiles;
        const int k3_0 = (int)(it - f * tiles) * ROWS;
        const __amdgpu_buffer_rsrc_t ri = frame_rsrc(scratch + f * nfft + (size_t)k3_0 * M, (unsigned)(16 * M * 8));
        cf v[16];
#pragma unroll
        for (int i = 0; i < C0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const v2f x = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, e0 * 8, scratch_index(0, (i + C0 * j) * T, M) * 8, SDRK_SCR_LD_AUX));
                v[i * R0 + j] = cf{x.x, x.y};
            }
        lds_fft_core<LOG2M, 1, SDRK_ROW_SYNC(T)>(v, lds, 0, rt, tw);
        __syncthreads();  // all rows are through their last LDS reads: the buffer becomes the transpose tile
        if (EPILOGUE == EPI_LOGPSD) {
            float* __restrict__ tile = reinterpret_cast<float*>(lds_all);  // [km][ROWS + 1]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[rev16(q)];
                tile[(rt + T * (q ^ xor_q)) * (ROWS + 1) + fr] = logpsd_db(z.x, z.y, eps);
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t ro = frame_rsrc(static_cast<float*>(out_raw) + f * nfft + k3_0,
                                                         (unsigned)((nfft - k3_0) * 4));
            // element e = tid + 16 T i of the ROWS x M tile: row r = e % ROWS (lanes), km = e / ROWS = tid / ROWS + T i
            const int r = tid & (ROWS - 1), km0 = tid / ROWS;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float val = tile[(km0 + T * i) * (ROWS + 1) + r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, (km0 * A + r) * 4,
                                                      i * T * A * 4, SDRK_ROW_ST_AUX);
            }
        } else {
            float2* __restrict__ tile = lds_all;  // [km][ROWS + 1]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[rev16(q)];
                tile[(rt + T * (q ^ xor_q)) * (ROWS + 1) + fr] = make_float2(z.x, z.y);
            }
            __syncthreads();
            const int r = tid & (ROWS - 1), km0 = tid / ROWS;
            if (EPILOGUE == EPI_COMPLEX) {
                // (buffer stores like the log branch: sixteen unrolled 64-bit addresses cost this branch its fourth wave per SIMD)
                const __amdgpu_buffer_rsrc_t ro = frame_rsrc(static_cast<float2*>(out_raw) + f * nfft + k3_0,
                                                             (unsigned)((nfft - k3_0) * 8));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float2 val = tile[(km0 + T * i) * (ROWS + 1) + r];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, val), ro, (km0 * A + r) * 8, i * T * A * 8, 0);
                }
            } else {
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const float2 val = tile[(km0 + T * i) * (ROWS + 1) + r];
                    epi_store_complex<EPILOGUE>(out_raw, f, nfft, (size_t)(km0 + T * i) * A + k3_0 + r, val, epi, eps);
                }
            }
        }
        __syncthreads();  // tile reads done before the next item's exchanges
    }
}

// row pass for M >= 1024 with the log epilogue (N = 2^19 ... 2^22).  A 16-row tile of M = 1024 (and an 8-row tile of
// M = 2048) fills the LDS; the generic kernel answers that with 512-thread workgroups of two 16-point sets per
// thread and a trickled register prefetch (66.5 us per 192 MiB chunk at M = 1024), and at M = 2048 its 8-row tiles
// leave in 32-byte pieces (115 us, 2.4 TB/s).  This kernel is the plain alternative that turned out faster: it walks
// 16-row bands as two 8-row HALVES, keeps the first half's dB values in registers (16 per thread) and writes both
// halves through one [M][17] float transpose tile — exactly the size of the 8-row exchange area — so that the
// stores are 64-byte pieces.  One 16-point set per thread, no prefetch, nothing to schedule by hand: at M = 1024 a
// half is 64 KiB, TWO 512-thread workgroups share a CU (four waves per SIMD) and overlap each other: 55.0 us
// (two sets per thread: 61.0); at M = 2048 one 1024-thread workgroup: 61.5 us (two sets: 65.8).  The complex
// epilogue (fft_c64) takes the same kernel: its 8-row halves are 64-byte pieces as they are.
template <int LOG2M, int EPILOGUE>
__global__ __launch_bounds__(LdsCfg<LOG2M>::T * 8, 4) void row_pass_pair_kernel(
    const float2* __restrict__ scratch, void* __restrict__ out_raw, size_t n_frames, int A,
    const float2* __restrict__ twM, float eps, int shift, EpiTab epi) {
    using C = LdsCfg<LOG2M>;
    constexpr int M = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0, WGT = T * 8;
    static_assert((size_t)M * 17 * sizeof(float) <= (size_t)8 * C::SLOT * sizeof(float2), "transpose tile must fit the exchange area");
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];   // 8 x SLOT; complex epilogue: at least [M][9] float2
    LdsTw<LOG2M> tw;
    lds_tw_init<LOG2M>(tw, twM, (int)threadIdx.x % T);
    const size_t nfft = (size_t)A * M;
    const int bands = A / 16;
    const size_t items = n_frames * (size_t)bands;
    const int xor_q = shift ? 8 : 0;
    for (size_t g = blockIdx.x; g < items; g += gridDim.x) {
        const size_t f = g / bands;
        const int k3_0 = (int)(g - f * bands) * 16;
        const __amdgpu_buffer_rsrc_t ri = frame_rsrc(scratch + f * nfft + (size_t)k3_0 * M, (unsigned)(16 * M * 8));
        // the thread's coordinates are re-derived per band from an opaque copy of its number: hoisted out of this
        // persistent loop, the dozen LDS / buffer offsets derived from them cost the registers that the first
        // half's 16 parked dB values then lack (128-VGPR cap at four waves per SIMD: 12 / 24 bytes of scratch)
        int tid_o = threadIdx.x;
        asm volatile("" : "+v"(tid_o));
        __builtin_assume(tid_o >= 0 && tid_o < WGT);
        const int fr = tid_o / T, rt = tid_o - fr * T, tid = tid_o;
        float2* __restrict__ lds = lds_all + (size_t)fr * C::SLOT;
        float val[2][16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            cf v[16];
            const int e0 = scratch_index(8 * h + fr, rt, M);
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const v2f x = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, e0 * 8, scratch_index(0, (i + C0 * j) * T, M) * 8, SDRK_SCR_LD_AUX));
                    v[i * R0 + j] = cf{x.x, x.y};
                }
            lds_fft_core<LOG2M, 1, SDRK_ROW_SYNC(T)>(v, lds, 0, rt, tw);
            if (EPILOGUE == EPI_LOGPSD) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const cf z = v[rev16(q)];
                    val[h][q] = logpsd_db(z.x, z.y, eps);
                }
            } else {
                // complex rows: 8 rows x 8 bytes are 64-byte pieces already; each half leaves through its own [M][9] tile
                __syncthreads();
                float2* __restrict__ tile = lds_all;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const cf z = v[rev16(q)];
                    tile[(rt + T * (q ^ xor_q)) * 9 + fr] = make_float2(z.x, z.y);
                }
                __syncthreads();
                if (EPILOGUE == EPI_COMPLEX) {
                    float2* __restrict__ o = static_cast<float2*>(out_raw) + f * nfft + k3_0 + 8 * h;
#pragma unroll 8
                    for (int i = 0; i < 8 * M / WGT; ++i) {
                        const int e = tid + WGT * i;
                        const int r = e & 7, km = e >> 3;
                        o[(size_t)km * A + r] = tile[km * 9 + r];
                    }
                } else {
#pragma unroll 4
                    for (int i = 0; i < 8 * M / WGT; ++i) {
                        const int e = tid + WGT * i;
                        const int r = e & 7, km = e >> 3;
                        epi_store_complex<EPILOGUE>(out_raw, f, nfft, (size_t)km * A + k3_0 + 8 * h + r, tile[km * 9 + r], epi, eps);
                    }
                }
                __syncthreads();  // tile reads done before the next half's exchanges
            }
        }
        if (EPILOGUE == EPI_LOGPSD) {
            __syncthreads();  // all rows are through their last LDS reads: the exchange area becomes the transpose tile
            float* __restrict__ tile = reinterpret_cast<float*>(lds_all);  // [km][17]
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 16; ++q) tile[(rt + T * (q ^ xor_q)) * 17 + 8 * h + fr] = val[h][q];
            __syncthreads();
            const __amdgpu_buffer_rsrc_t ro = frame_rsrc(static_cast<float*>(out_raw) + f * nfft + k3_0, (unsigned)((nfft - k3_0) * 4));
            const int r = tid & 15, km0 = tid >> 4;                              // km = km0 + (WGT / 16) i
#pragma unroll 8
            for (int i = 0; i < 16 * M / WGT; ++i) {
                const float x = tile[(km0 + (WGT / 16) * i) * 17 + r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), ro, (km0 * A + r) * 4, i * (WGT / 16) * A * 4, SDRK_ROW_ST_AUX);
            }
            __syncthreads();  // tile reads done before the next band's exchanges
        }
    }
}

// The M = 2048 row pass with ONE WAVE PER ROW (round 4; N = 2^20 ... 2^22, log epilogue).
//
// lds_fft_core gives a row to T = M / 16 = 128 threads — two waves — and puts a workgroup barrier around every exchange.
// In row_pass_pair_kernel<11> that marches all sixteen waves of the one 1024-thread workgroup a CU has room for through
// load -> transform -> transpose -> store in lockstep: a wave parked at s_waitcnt / s_barrier 40 % of its cycles, VALU
// 51 % of a SIMD, an L2 channel busy only 63 % of the dispatch — 70 us per 24-frame chunk where its bytes need 47
// (profiles/r04/cfg5_pair_kernel_sq_activity.txt, cfg5_pair_kernel_mem_counters.txt).  Here each LANE is two of those 128 threads (tau = lane and
// lane + 64: 32 points, 64 registers), so that
//   * a row's exchanges stay inside one wave and need NO barrier: the wave's own LDS queue is the ordering;
//   * a workgroup is four waves = four rows, 70 KiB of LDS, and TWO workgroups share a CU and overlap each other's
//     phases — at two waves per SIMD, i.e. 256 registers per wave, which is what lets the dB values of three earlier
//     four-row steps (3 x 32) wait in registers until the sixteen-row band leaves in 64-byte pieces (the 128-register
//     forms of this idea needed ~35 registers of spill, and 8-row bands leave in 32-byte pieces: both measured, A.11);
//   * barriers remain only around the band's transposed store ([M / 2][17] floats: exactly the exchange area, twice).
// Measured on config 5 (tools/ab_cfg.py, rocprofv3 kernel trace): 69.8 -> 57.6 us per chunk, N = 2^20 1.49 -> 1.40 ms;
// N = 2^21 / 2^22 packed frames -3 ... -6 %.  Rows equal the pair kernel's to the last bits that the second thread's
// derived table entries leave (W_N^(tau + 64 + T i) is formed as W_N^(tau + T i) W_N^64: one rounding).
// HR rows (= waves) per step, 16 / HR steps per band, the band's rows leaving through KH = 8 / HR slices of km.
// MIP: besides the rows, write each band's max over its 16 rows per km — 16 CONSECUTIVE bins of the output row, since
// k = A km + k3 — i.e. the row max-hold-decimated by 16, as mip[frame][band = k3 / 16][km (shifted like the row)]: N / 16
// floats per frame, 8 KiB contiguous per band.  The waterfall's decimated read-out (sdrk_waterfall_read_decimated, any
// factor that is a multiple of 16) then reads 1/16 of the bytes instead of every 4 MiB row again.
template <int LOG2M, int HR, bool MIP>
__global__ __launch_bounds__(64 * HR, 2) void row_pass_wave_kernel(
    const float2* __restrict__ scratch, float* __restrict__ out, size_t n_frames, int A,
    const float2* __restrict__ twM, float eps, int shift, float* __restrict__ mip) {
    using C = LdsCfg<LOG2M>;
    constexpr int M = C::N, R0 = C::R0, T = C::T, C0 = 16 / R0, WGT = 64 * HR, STEPS = 16 / HR, KH = 8 / HR, MK = M / KH;
    static_assert(T == 128 && (HR == 4 || HR == 8), "one wave per row of 2048 points; 4 or 8 rows per step");
    static_assert((size_t)MK * 17 * sizeof(float) <= (size_t)HR * C::SLOT * sizeof(float2), "transpose slice must fit the exchange area");
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // Table entries of the lane's first thread only; those of its second thread, tau + 64, follow from them:
    // W_N^(tau + 64 + T i) = W_N^(tau + T i) W_N^64, and the later passes' W_{N_p}^(tau % M_p) are the same for both
    // (M_p divides 64) — ten registers less.  They are WAITED FOR here: a register still pending on the loop's entry edge
    // makes the compiler put a vmcnt(0) in front of its first use inside the loop — in the middle of a transform, where
    // it would wait for the previous band's stores.
    LdsTw<LOG2M> twa;
    lds_tw_init<LOG2M>(twa, twM, (int)threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < 16 / R0; ++i) asm volatile("" :: "v"(twa.w0[i].x), "v"(twa.w0[i].y));
#pragma unroll
    for (int i = 0; i < C::P; ++i) asm volatile("" :: "v"(twa.wp[i].x), "v"(twa.wp[i].y));
    float2* __restrict__ lds = lds_all + (size_t)wave * C::SLOT;
    const size_t nfft = (size_t)A * M;
    const int bands = A / 16;
    const size_t items = n_frames * (size_t)bands;
    const int xor_q = shift ? 8 : 0;
    for (size_t g = blockIdx.x; g < items; g += gridDim.x) {
        const size_t f = g / bands;
        const int k3_0 = (int)(g - f * bands) * 16;
        const __amdgpu_buffer_rsrc_t ri = frame_rsrc(scratch + f * nfft + (size_t)k3_0 * M, (unsigned)(16 * M * 8));
        float val[STEPS][2][16];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            // (the thread's coordinates are re-derived wherever they are used, from an opaque copy of its number: hoisted
            // out of the persistent loop, the LDS / buffer offsets derived from them would be live across everything)
            int tid_o = threadIdx.x;
            asm volatile("" : "+v"(tid_o));
            __builtin_assume(tid_o >= 0 && tid_o < WGT);
            const int taua = tid_o & 63, taub = taua + 64;
            cf va[16], vb[16];
            const int ea = scratch_index(HR * st + wave, taua, M), eb = scratch_index(HR * st + wave, taub, M);
#pragma unroll
            for (int i = 0; i < C0; ++i)
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const int u = scratch_index(0, (i + C0 * j) * T, M) * 8;
                    const v2f xa = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, ea * 8, u, SDRK_SCR_LD_AUX));
                    const v2f xb = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, eb * 8, u, SDRK_SCR_LD_AUX));
                    va[i * R0 + j] = cf{xa.x, xa.y};
                    vb[i * R0 + j] = cf{xb.x, xb.y};
                }
            const LdsTw<LOG2M> twb = nv2_tw_b<LOG2M>(twa);
            lds_fft_core_nv2<LOG2M>(va, vb, lds, taua, taub, twa, twb);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf za = va[rev16(q)], zb = vb[rev16(q)];
                val[st][0][q] = logpsd_db(za.x, za.y, eps);
                val[st][1][q] = logpsd_db(zb.x, zb.y, eps);
            }
        }
        if (MIP) {
            // partial maxima of this wave's STEPS rows, one float per (thread of the transform, q): part[wave][j = 16 h + q][lane]
            static_assert((size_t)HR * 32 * 64 * sizeof(float) <= (size_t)HR * C::SLOT * sizeof(float2), "partial maxima must fit the exchange area");
            float* __restrict__ part = reinterpret_cast<float*>(lds_all);
            __syncthreads();  // every wave is through its last exchange reads
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            __builtin_assume(tid >= 0 && tid < WGT);
            const int lane = tid & 63;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float m = val[0][h][q];
#pragma unroll
                    for (int st = 1; st < STEPS; ++st) m = fmaxf(m, val[st][h][q]);
                    part[(wave * 32 + 16 * h + q) * 64 + lane] = m;
                }
            __syncthreads();
            float* __restrict__ mrow = mip + (f * bands + (size_t)(k3_0 >> 4)) * M;
#pragma unroll
            for (int i = 0; i < 2048 / WGT; ++i) {
                const int s = tid + WGT * i, j = s >> 6;                       // j = 16 h + q (uniform per wave), lane = s & 63
                float m = part[j * 64 + lane];
#pragma unroll
                for (int w = 1; w < HR; ++w) m = fmaxf(m, part[(w * 32 + j) * 64 + lane]);
                const int km = lane + 64 * (j >> 4) + T * ((j & 15) ^ xor_q);   // tau + T (q ^ xor_q), tau = lane + 64 h
                mrow[km] = m;
            }
        }
        float* __restrict__ tile = reinterpret_cast<float*>(lds_all);  // [km - kh MK][17]
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            __syncthreads();  // the exchange area (or the previous slice of the tile) is free
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            __builtin_assume(tid >= 0 && tid < WGT);
            {
                const int taua = tid & 63, taub = taua + 64;
#pragma unroll
                for (int st = 0; st < STEPS; ++st)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int qq = q ^ xor_q;                          // km = tau + T qq; slice kh holds qq / (16 / KH) == kh
                        if (qq / (16 / KH) == kh) {
                            tile[(taua + T * (qq % (16 / KH))) * 17 + HR * st + wave] = val[st][0][q];
                            tile[(taub + T * (qq % (16 / KH))) * 17 + HR * st + wave] = val[st][1][q];
                        }
                    }
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t ro = frame_rsrc(out + f * nfft + k3_0, (unsigned)((nfft - k3_0) * 4));
            const int r = tid & 15, km0 = tid >> 4;                              // km = kh MK + km0 + (WGT / 16) i
#pragma unroll 8
            for (int i = 0; i < 16 * MK / WGT; ++i) {
                const float x = tile[(km0 + (WGT / 16) * i) * 17 + r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), ro, ((kh * MK + km0) * A + r) * 4,
                                                      i * (WGT / 16) * A * 4, SDRK_ROW_ST_AUX);
            }
        }
        __syncthreads();  // tile reads done before the next band's exchanges
    }
}

// frame lengths whose row pass (row_pass_wave_kernel: M = 2048, log epilogue) can also write the max-hold-by-16 rows
bool fft_tiled2_has_mip(int nfft, int epilogue) {
#ifdef SDRK_ROW_PAIR_2048
    return false;
#else
    return epilogue == EPI_LOGPSD && (nfft == (1 << 20) || nfft == (1 << 21) || nfft == (1 << 22));
#endif
}

bool fft_tiled2_split(int nfft, int* log2a, int* log2m) {
    int lg = 0;
    while ((1 << lg) < nfft) ++lg;
    if ((1 << lg) != nfft || lg < 15 || lg > 22) return false;
    int la = lg / 2 < 7 ? 7 : lg / 2;   // 2^15 -> 128 x 256 ; 2^17 -> 256 x 512 ; 2^21 -> 1024 x 2048 ; 2^22 -> 2048 x 2048
    // 2^20 as 512 x 2048 rather than 1024 x 1024: the A = 512 col pass fetches whole 128-byte lines (80 us per
    // 192 MiB chunk against 97 for the 8-column tiles of A = 1024) and the M = 2048 row pass costs 68 against 56:
    // 1.46 against 1.54 ms per 256 frames, the same in three back-to-back comparisons
    if (lg == 20) la = 9;
    *log2a = la;
    *log2m = lg - la;
    return true;
}

// col pass through col_pass_staged_kernel: one 512-thread workgroup per CU (133.5 KiB of LDS)
// bytes of each input frame that exist (LaunchArgs::in_valid; the whole frame by default)
static unsigned col_in_bytes(const LaunchArgs& a) { return (unsigned)((a.in_valid ? a.in_valid : (size_t)a.nfft) * 8); }

template <int LOG2A, int W>
static hipError_t launch_col_staged(const LaunchArgs& a, const float2* src, size_t nf, int M) {
    using C = LdsCfg<LOG2A>;
    const unsigned tiles = (unsigned)(M / W);
    unsigned grid = (unsigned)a.num_cus * staged_per_cu(C::SLOT, C::N, W);
    if (grid >= tiles) grid -= grid % tiles;             // whole runs of frames per tile position
    const size_t lds_bytes = ((size_t)C::SLOT * W + (size_t)C::N * W) * sizeof(float2);
    const float2* twA = static_cast<const float2*>(a.d_twiddle_2p);
    const float2* t1T = twA + 2048 + 2048;
    const float2* t2 = t1T + (size_t)(C::T) * M;
    float2* scratch = static_cast<float2*>(a.d_scratch);
#define SDRK_COLS(WIN)                                                                                           \
    do {                                                                                                         \
        auto kern = col_pass_staged_kernel<LOG2A, WIN, W>;                                                       \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                     \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);              \
        if (e0 != hipSuccess) return e0;                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::T * W), lds_bytes, a.stream, src, a.frame_stride, scratch, nf, M, \
                           a.d_window, twA, t1T, t2, col_in_bytes(a));                                           \
    } while (0)
    if (a.d_window) SDRK_COLS(true); else SDRK_COLS(false);
#undef SDRK_COLS
    return hipGetLastError();
}

template <int LOG2A>
static hipError_t launch_col_tiles(const LaunchArgs& a, const float2* src, size_t nf, int M, unsigned grid_cap) {
    using C = LdsCfg<LOG2A>;
    constexpr int W = COL_TILE_W(LOG2A);
    const size_t lds_bytes = (size_t)(C::SLOT) * W * sizeof(float2);
    const unsigned tiles = (unsigned)(M / W);
    const size_t items = nf * (size_t)tiles;
    // The software-pipelined (FIXED) instances hold two tiles and their per-position factors in registers
    // (<= 168 VGPRs): 3 workgroups per CU are resident, and a persistent grid must not exceed what is resident.
    constexpr bool CAN_FIX = col_can_fix(C::T, W);
    if (W == 8 && !CAN_FIX) grid_cap = (unsigned)a.num_cus * 2;
    if (CAN_FIX && grid_cap > (unsigned)a.num_cus * 3) grid_cap = (unsigned)a.num_cus * 3;
    unsigned grid = (unsigned)(items < grid_cap ? items : grid_cap);
    if (W == 8 && grid >= 16) grid &= ~15u;
    bool fixed = false;
    if (CAN_FIX && grid >= tiles) {          // items is a multiple of tiles, so a rounded-down grid still divides evenly
        grid -= grid % tiles;
        fixed = true;
    }
    const float2* twA = static_cast<const float2*>(a.d_twiddle_2p);
    const float2* t1T = twA + 2048 + 2048;
    const float2* t2 = t1T + (size_t)(C::T) * M;
    float2* scratch = static_cast<float2*>(a.d_scratch);
    // overlapped frames: rows shared between consecutive frames (see the kernel's SH)
    int sh = 16;
    if (CAN_FIX && fixed && a.frame_stride % (size_t)M == 0) {
        const size_t rows = a.frame_stride / (size_t)M;
        if (rows == (size_t)C::T * 8) sh = 8;
        else if (rows == (size_t)C::T * 4) sh = 4;
    }
#define SDRK_COL(WIN, FIX, SHV)                                                                                  \
    do {                                                                                                         \
        auto kern = col_pass_kernel<LOG2A, WIN, W, FIX, SHV>;                                                    \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                     \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);              \
        if (e0 != hipSuccess) return e0;                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::T * W), lds_bytes, a.stream, src, a.frame_stride, scratch, nf, \
                           M, a.d_window, twA, t1T, t2, col_in_bytes(a));                                           \
    } while (0)
    if (CAN_FIX && fixed) {
        if (sh == 8) { if (a.d_window) SDRK_COL(true, CAN_FIX, (CAN_FIX ? 8 : 16)); else SDRK_COL(false, CAN_FIX, (CAN_FIX ? 8 : 16)); }
        else if (sh == 4) { if (a.d_window) SDRK_COL(true, CAN_FIX, (CAN_FIX ? 4 : 16)); else SDRK_COL(false, CAN_FIX, (CAN_FIX ? 4 : 16)); }
        else { if (a.d_window) SDRK_COL(true, CAN_FIX, 16); else SDRK_COL(false, CAN_FIX, 16); }
    } else {
        if (a.d_window) SDRK_COL(true, false, 16); else SDRK_COL(false, false, 16);
    }
#undef SDRK_COL
    return hipGetLastError();
}

template <int LOG2A>
static hipError_t launch_col(const LaunchArgs& a, const float2* src, size_t nf, int M, unsigned grid_cap) {
    if constexpr (STAGED_W(LOG2A) != 0) return launch_col_staged<LOG2A, STAGED_W(LOG2A)>(a, src, nf, M);
    else return launch_col_tiles<LOG2A>(a, src, nf, M, grid_cap);
}


// row pass through row_pass_pair_kernel (M >= 1024): as many workgroups per CU as their LDS allows
static EpiTab epi_tab(const LaunchArgs& a) {
    EpiTab t;
    t.tab = a.d_epi_tab;
    t.n_out = a.epi_n_out;
    t.rot = a.shift ? a.epi_n_out / 2 : 0;
    t.inv_m = 1.0f / (float)a.nfft;
    return t;
}
// element size and frame stride (in elements) of a row pass's output, by epilogue
static size_t out_elem_bytes(const LaunchArgs& a) { return (a.epilogue == EPI_LOGPSD || a.epilogue == EPI_BLU_POST_LOG) ? sizeof(float) : sizeof(float2); }
static size_t out_row_elems(const LaunchArgs& a) { return (a.epilogue == EPI_BLU_POST_LOG || a.epilogue == EPI_BLU_POST_C64) ? (size_t)a.epi_n_out : (size_t)a.nfft; }

template <int LOG2M>
static hipError_t launch_row_pair(const LaunchArgs& a, void* dst, size_t nf, int A, float* mip) {
    using C = LdsCfg<LOG2M>;
    const size_t items = nf * (size_t)(A / 16);
    const size_t xch = (size_t)8 * C::SLOT * sizeof(float2), ctile = (size_t)C::N * 9 * sizeof(float2);
    const size_t lds_bytes = a.epilogue == EPI_LOGPSD ? xch : (xch > ctile ? xch : ctile);
    const size_t per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
    const size_t cap = (size_t)a.num_cus * per_cu;
    const unsigned grid = (unsigned)(items < cap ? items : cap);
    const float2* twM = static_cast<const float2*>(a.d_twiddle_2p) + 2048;
    // M = 2048 with the log epilogue: one wave per row, two four-wave workgroups per CU (row_pass_wave_kernel above);
    // SDRK_ROW_PAIR_2048 (A/B builds) keeps the one 1024-thread workgroup of row_pass_pair_kernel
#ifndef SDRK_ROW_PAIR_2048
    if constexpr (LOG2M == 11) {
        if (a.epilogue == EPI_LOGPSD) {
            constexpr int HR = 4;
            const size_t lds_h = (size_t)HR * C::SLOT * sizeof(float2);
            const size_t caph = (size_t)a.num_cus * (8 / HR);
            const unsigned gh = (unsigned)(items < caph ? items : caph);
#define SDRK_ROWW(MIPV)                                                                                          \
    do {                                                                                                         \
        auto kernh = row_pass_wave_kernel<LOG2M, HR, MIPV>;                                                      \
        static std::atomic<uint64_t> lds_ok_h{0};                                                                \
        hipError_t eh = ensure_dynamic_lds(reinterpret_cast<const void*>(kernh), lds_h, lds_ok_h);               \
        if (eh != hipSuccess) return eh;                                                                         \
        hipLaunchKernelGGL(kernh, dim3(gh), dim3(64 * HR), lds_h, a.stream, static_cast<const float2*>(a.d_scratch), \
                           static_cast<float*>(dst), nf, A, twM, a.eps, a.shift, mip);                           \
    } while (0)
            if (mip) SDRK_ROWW(true); else SDRK_ROWW(false);
#undef SDRK_ROWW
            return hipGetLastError();
        }
    }
#endif
#define SDRK_ROWP(E)                                                                                             \
    do {                                                                                                         \
        auto kern = row_pass_pair_kernel<LOG2M, E>;                                                              \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                     \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);              \
        if (e0 != hipSuccess) return e0;                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::T * 8), lds_bytes, a.stream, static_cast<const float2*>(a.d_scratch), dst, nf, \
                           A, twM, a.eps, a.epilogue >= EPI_BLU_MUL ? 0 : a.shift, epi_tab(a));                   \
    } while (0)
    switch (a.epilogue) {
        case EPI_LOGPSD: SDRK_ROWP(EPI_LOGPSD); break;
        case EPI_BLU_MUL: SDRK_ROWP(EPI_BLU_MUL); break;
        case EPI_BLU_POST_LOG: SDRK_ROWP(EPI_BLU_POST_LOG); break;
        case EPI_BLU_POST_C64: SDRK_ROWP(EPI_BLU_POST_C64); break;
        default: SDRK_ROWP(EPI_COMPLEX); break;
    }
#undef SDRK_ROWP
    return hipGetLastError();
}

template <int LOG2M>
static hipError_t launch_row(const LaunchArgs& a, void* dst, size_t nf, int A, unsigned grid_cap, float* mip) {
    using C = LdsCfg<LOG2M>;
    if constexpr (LOG2M >= 10) {
        return launch_row_pair<LOG2M>(a, dst, nf, A, mip);
    } else {
        constexpr int ROWS = 16;
        // LDS: the exchange area (ROWS x 17/16 M complex) or the complex transpose tile (M x (ROWS+1)), whichever is larger
        const size_t xch = (size_t)ROWS * C::SLOT, tile = (size_t)C::N * (ROWS + 1);
        const size_t lds_bytes = (xch > tile ? xch : tile) * sizeof(float2);
        const size_t items = nf * (size_t)(A / ROWS);
        const unsigned grid = (unsigned)(items < grid_cap ? items : grid_cap);
        const float2* twM = static_cast<const float2*>(a.d_twiddle_2p) + 2048;
        const float2* scratch = static_cast<const float2*>(a.d_scratch);
#define SDRK_ROW(E)                                                                                              \
    do {                                                                                                         \
        auto kern = row_pass_kernel<LOG2M, E>;                                                                   \
        static std::atomic<uint64_t> lds_ok{0};   /* per instantiation, one bit per device */                     \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);              \
        if (e0 != hipSuccess) return e0;                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::T * ROWS), lds_bytes, a.stream, scratch, dst, nf, A, twM, a.eps, \
                           a.epilogue >= EPI_BLU_MUL ? 0 : a.shift, epi_tab(a));   /* EPI_BLU_*: the transform itself runs unshifted */ \
    } while (0)
        switch (a.epilogue) {
            case EPI_LOGPSD: SDRK_ROW(EPI_LOGPSD); break;
            case EPI_BLU_MUL: SDRK_ROW(EPI_BLU_MUL); break;
            case EPI_BLU_POST_LOG: SDRK_ROW(EPI_BLU_POST_LOG); break;
            case EPI_BLU_POST_C64: SDRK_ROW(EPI_BLU_POST_C64); break;
            default: SDRK_ROW(EPI_COMPLEX); break;
        }
#undef SDRK_ROW
        return hipGetLastError();
    }
}

static hipError_t launch_fft_tiled2_serial(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    int la = 0, lm = 0;
    if (!fft_tiled2_split(a.nfft, &la, &lm)) return hipErrorInvalidValue;
    const int A = 1 << la, M = 1 << lm;
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const size_t out_elem = out_elem_bytes(a), out_row = out_row_elems(a);
    // workgroups per CU: LDS 136 B per point of A (or M); at most 2048 threads
    auto cap = [&](int L) {
        size_t per_cu = (160 * 1024) / ((size_t)136 * L);
        const size_t by_threads = 2048 / L;
        if (per_cu > by_threads) per_cu = by_threads;
        if (per_cu > 4) per_cu = 4;
        if (per_cu < 1) per_cu = 1;
        return (unsigned)(a.num_cus * per_cu);
    };
    // Frames per chunk: the scratch capacity, rounded down to a whole number of grid passes of BOTH kernels
    // (col: 3 workgroups per CU when software-pipelined, else cap(A); row: cap(M)) so that no persistent
    // workgroup gets one work item more than its neighbours (N = 65536: 256 frames = 5.33 col items per
    // workgroup became 192 = exactly 4 col and 3 row items).
    size_t chunk = a.scratch_frames;
    {
        auto gcd = [](size_t x, size_t y) { while (y) { size_t t = x % y; x = y; y = t; } return x; };
        const int Wc = COL_TILE_W(la <= 11 ? la : 11), Rr = 16;
        const bool pipelined = col_can_fix(A / 16, Wc);
        size_t col_grid = STAGED_W(la) ? (size_t)a.num_cus * staged_per_cu(A + A / 16, A, Wc)
                        : (pipelined ? (size_t)a.num_cus * 3 : (Wc == 8 ? (size_t)a.num_cus * 2 : cap(A < 1024 ? A : 1024)));
        if (pipelined && col_grid > cap(A)) col_grid = cap(A);
        const size_t row_grid = M >= 2048 ? (size_t)a.num_cus : (M == 1024 ? (size_t)a.num_cus * 2 : cap(M));
        size_t cw = col_grid / (size_t)(M / Wc), rw = row_grid / (size_t)(A / Rr);
        if (cw < 1) cw = 1;
        if (rw < 1) rw = 1;
        const size_t l = cw / gcd(cw, rw) * rw;
        if (chunk >= l) chunk -= chunk % l;
    }
    for (size_t f0 = 0; f0 < a.n_frames; f0 += chunk) {
        const size_t nf = (a.n_frames - f0 < chunk) ? a.n_frames - f0 : chunk;
        const float2* src = iq + f0 * a.frame_stride;
        void* dst = static_cast<char*>(a.d_out) + f0 * out_row * out_elem;
        hipError_t e;
        switch (la) {
            case 7: e = launch_col<7>(a, src, nf, M, cap(A)); break;
            case 8: e = launch_col<8>(a, src, nf, M, cap(A)); break;
            case 9: e = launch_col<9>(a, src, nf, M, cap(A)); break;
            case 10: e = launch_col<10>(a, src, nf, M, cap(A)); break;
            default: e = launch_col<11>(a, src, nf, M, cap(1024)); break;
        }
        if (e != hipSuccess) return e;
        float* mip = (a.d_mip && fft_tiled2_has_mip(a.nfft, a.epilogue)) ? a.d_mip + f0 * (size_t)(a.nfft / 16) : nullptr;
            if (a.mip_written) *a.mip_written = mip != nullptr;
        switch (lm) {
            case 8: e = launch_row<8>(a, dst, nf, A, cap(M), nullptr); break;
            case 9: e = launch_row<9>(a, dst, nf, A, cap(M), nullptr); break;
            case 10: e = launch_row<10>(a, dst, nf, A, cap(M), nullptr); break;
            default: e = launch_row<11>(a, dst, nf, A, cap(1024), mip); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One chunk's col pass / row pass (the kernels are chosen by the split alone).
static hipError_t col_pass(const LaunchArgs& a, int la, const float2* src, size_t nf, int M, unsigned grid_cap, unsigned grid_cap_1024) {
    switch (la) {
        case 7: return launch_col<7>(a, src, nf, M, grid_cap);
        case 8: return launch_col<8>(a, src, nf, M, grid_cap);
        case 9: return launch_col<9>(a, src, nf, M, grid_cap);
        case 10: return launch_col<10>(a, src, nf, M, grid_cap);
        default: return launch_col<11>(a, src, nf, M, grid_cap_1024);
    }
}
static hipError_t row_pass(const LaunchArgs& a, int lm, void* dst, size_t nf, int A, unsigned grid_cap, unsigned grid_cap_1024, float* mip) {
    switch (lm) {
        case 8: return launch_row<8>(a, dst, nf, A, grid_cap, nullptr);
        case 9: return launch_row<9>(a, dst, nf, A, grid_cap, nullptr);
        case 10: return launch_row<10>(a, dst, nf, A, grid_cap, nullptr);
        default: return launch_row<11>(a, dst, nf, A, grid_cap_1024, mip);
    }
}

hipError_t launch_fft_tiled2(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    int la = 0, lm = 0;
    if (!fft_tiled2_split(a.nfft, &la, &lm)) return hipErrorInvalidValue;
    const int A = 1 << la, M = 1 << lm;
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const size_t out_elem = out_elem_bytes(a), out_row = out_row_elems(a);
    // Overlapped form: two role-sized grids resident together, two scratch halves.  Needs the second stream and
    // its events, and at least three half-chunks of work (otherwise nothing overlaps).
    const bool want_ovl = a.stream2 && a.ev_fork && a.col_cus > 0 && a.row_cus > 0;
    const int ccus = want_ovl ? a.col_cus : a.num_cus, rcus = want_ovl ? a.row_cus : a.num_cus;
    // workgroups per CU: LDS 136 B per point of A (or M); at most 2048 threads
    auto cap = [&](int L, int cus) {
        size_t per_cu = (160 * 1024) / ((size_t)136 * L);
        const size_t by_threads = 2048 / L;
        if (per_cu > by_threads) per_cu = by_threads;
        if (per_cu > 4) per_cu = 4;
        if (per_cu < 1) per_cu = 1;
        return (unsigned)(cus * per_cu);
    };
    // Frames per chunk: the scratch capacity, rounded down to a whole number of grid passes of BOTH kernels
    // (col: 3 workgroups per CU when software-pipelined, else cap(A); row: cap(M)) so that no persistent
    // workgroup gets one work item more than its neighbours (N = 65536: 256 frames = 5.33 col items per
    // workgroup became 192 = exactly 4 col and 3 row items).
    auto round_chunk = [&](size_t chunk) {
        auto gcd = [](size_t x, size_t y) { while (y) { size_t t = x % y; x = y; y = t; } return x; };
        const int Wc = COL_TILE_W(la <= 11 ? la : 11), Rr = 16;
        const bool pipelined = col_can_fix(A / 16, Wc);
        size_t col_grid = STAGED_W(la) ? (size_t)ccus * staged_per_cu(A + A / 16, A, Wc)
                        : (pipelined ? (size_t)ccus * 3 : (Wc == 8 ? (size_t)ccus * 2 : cap(A < 1024 ? A : 1024, ccus)));
        if (pipelined && col_grid > cap(A, ccus)) col_grid = cap(A, ccus);
        const size_t row_grid = M >= 2048 ? (size_t)rcus : (M == 1024 ? (size_t)rcus * 2 : cap(M, rcus));
        size_t cw = col_grid / (size_t)(M / Wc), rw = row_grid / (size_t)(A / Rr);
        if (cw < 1) cw = 1;
        if (rw < 1) rw = 1;
        const size_t l = cw / gcd(cw, rw) * rw;
        if (chunk >= l) chunk -= chunk % l;
        return chunk;
    };
    LaunchArgs ca = a, ra = a;
    ca.num_cus = ccus;
    ra.num_cus = rcus;
    const size_t half = round_chunk(a.scratch_frames / 2);
    if (want_ovl && half >= 1 && a.n_frames > 2 * half) {
        // col(i) on `stream` into scratch half i % 2; row(i) on `stream2` out of it, beside col(i + 1).  ev_col[h]:
        // half h written; ev_row[h]: half h read (free for col(i + 2)).  Everything is joined back into `stream`
        // at the end, so the caller's stream order (and its timing events) cover the whole transform.
        ra.stream = a.stream2;
        hipError_t e = hipEventRecord(a.ev_fork, a.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(a.stream2, a.ev_fork, 0);
        if (e != hipSuccess) return e;
        size_t i = 0;
        for (size_t f0 = 0; f0 < a.n_frames; f0 += half, ++i) {
            const size_t nf = (a.n_frames - f0 < half) ? a.n_frames - f0 : half;
            const int h = (int)(i & 1);
            ca.d_scratch = ra.d_scratch = static_cast<float2*>(a.d_scratch) + (size_t)h * half * (size_t)a.nfft;
            if (i >= 2 && (e = hipStreamWaitEvent(a.stream, a.ev_row[h], 0)) != hipSuccess) return e;
            e = col_pass(ca, la, iq + f0 * a.frame_stride, nf, M, cap(A, ccus), cap(1024, ccus));
            if (e == hipSuccess) e = hipEventRecord(a.ev_col[h], a.stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(a.stream2, a.ev_col[h], 0);
            if (e != hipSuccess) return e;
            float* mip = (a.d_mip && fft_tiled2_has_mip(a.nfft, a.epilogue)) ? a.d_mip + f0 * (size_t)(a.nfft / 16) : nullptr;
            if (a.mip_written) *a.mip_written = mip != nullptr;
            e = row_pass(ra, lm, static_cast<char*>(a.d_out) + f0 * out_row * out_elem, nf, A, cap(M, rcus), cap(1024, rcus), mip);
            if (e == hipSuccess) e = hipEventRecord(a.ev_row[h], a.stream2);
            if (e != hipSuccess) return e;
        }
        e = hipStreamWaitEvent(a.stream, a.ev_row[0], 0);
        if (e == hipSuccess) e = hipStreamWaitEvent(a.stream, a.ev_row[1], 0);
        return e;
    }
    return launch_fft_tiled2_serial(a);   // one stream, whole-device grids, the whole scratch per chunk
}

}  // namespace sdrk
