// fft4096_wave.hip — EXPERIMENT (round 5), NOT part of the library (kept beside the probes; it was built into a variant with
// `fft4096_wave.hip` in the Makefile's SRCS and `launch_fft4096` forwarding to `launch_fft4096_wave`, git history `r05_exp19`).
// Result: parity-green (142 GPU tests, 3.5e-6) and 26 % SLOWER than the workgroup-per-frame flagship (11.4 against 9.0 ms per 2^20
// frames, the same on every buffer pairing: compute-bound) — with one wave per SIMD nothing overlaps the wave's own VALU, LDS
// and memory-instruction issue; they add up to 22 800 cycles per frame where the estimate for overlapped issue was 10 500.
// The N = 4096 transform with ONE WAVE per frame.  Each lane plays four of the
// flagship's threads (tau = lane + 64 s, s = 0..3: 64 points in 128 registers, the next frame's 64 loads parked in 128 more),
// the exchanges go through the wave's own 34.8 KiB LDS slot and need no workgroup barrier (the wave's LDS queue is the order),
// four such waves per CU (one per SIMD).  A wave reads its frame as 64 consecutive 512-byte pieces and writes its row as 64
// consecutive 256-byte pieces: the no-arithmetic copy of that shape streams 5 % faster than the workgroup-per-frame one
// (tools/phaseprobe.hip vave).  Same decomposition as fft_lds_core.h for N = 4096 = 16 x 16 x 16 (twiddles by power tree).
#include "../../sdr-iq-visualizer_amd/csrc/fft_lds_core.h"

namespace sdrk {

__device__ __forceinline__ void wave_sync() {
    SDRK_WAVE_SYNC_FENCE();
    __builtin_amdgcn_wave_barrier();
    SDRK_WAVE_SYNC_FENCE();
}

template <bool HAS_WINDOW, int EPILOGUE>
__global__ __launch_bounds__(256, 1) void fft4096_wave_kernel(
    const float2* __restrict__ iq, size_t frame_stride, void* __restrict__ out_raw, size_t n_frames,
    const float* __restrict__ window, const float2* __restrict__ twN, float eps, int shift) {
    using C = LdsCfg<12>;
    constexpr int NV = 4;
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* __restrict__ lds = lds_all + (size_t)wave * C::SLOT;
    float* __restrict__ lds_win = reinterpret_cast<float*>(lds_all + (size_t)4 * C::SLOT);
    if (HAS_WINDOW) {
#pragma unroll
        for (int j = 0; j < 16; ++j) lds_win[threadIdx.x + 256 * j] = window[threadIdx.x + 256 * j];
        __syncthreads();
    }
    // twiddle bases: pass 0 needs W_4096^tau per set, pass 1 W_256^(tau % 16) (the same for the four sets: 64 s = 0 mod 16)
    cf w0[NV];
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        const float2 t = twN[lane + 64 * s];
        w0[s] = cf{t.x, t.y};
    }
    const float2 t1 = twN[(lane & 15) * 16];
    const cf wp1 = cf{t1.x, t1.y};
    const int xor_q = shift ? 8 : 0;
    constexpr int OUT_ELEM = (EPILOGUE == EPI_LOGPSD ? 4 : 8);

    const size_t first = (size_t)blockIdx.x * 4 + wave, step = (size_t)gridDim.x * 4;
    v2u nxt[64];
    auto issue = [&](size_t fr) {
        if (fr >= n_frames) fr = first;   // harmless re-read past the end
        __amdgpu_buffer_rsrc_t r = frame_rsrc(iq + fr * frame_stride, 4096 * 8);
#pragma unroll
        for (int j = 0; j < 64; ++j) nxt[j] = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, j * 512, 2);
    };
    if (first < n_frames) issue(first);
    for (size_t f = first; f < n_frames; f += step) {
        cf v[NV][16];
        // x[tau_s + 256 jj] = element lane + 64 (s + 4 jj)
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const v2f t = __builtin_bit_cast(v2f, nxt[s + 4 * jj]);
                v[s][jj] = cf{t.x, t.y};
            }
        issue(f + step);
        // ---- pass 0: DFT-16 over the stride-256 points, times W_4096^(tau k) ----
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            if (HAS_WINDOW) {
                float win[16];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) win[jj] = lds_win[lane + 64 * s + 256 * jj];
                radix16<true>(v[s], win);
            } else {
                radix16(v[s]);
            }
            cf w[16], w1 = w0[s];
            asm volatile("" : "+v"(w1.x), "+v"(w1.y));   // keep the tree inside the frame loop
            pow_tree(w1, w);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[s][rev16(k)] = cmul(v[s][rev16(k)], w[k]);
            __builtin_amdgcn_sched_barrier(0);            // one set at a time
        }
        wave_sync();   // the previous frame's last-pass reads are done
        constexpr int S1 = C::Mp(0) + C::pad(1);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const cf z = v[s][rev16(k)];
                lds[(lane + 64 * s) + S1 * k] = make_float2(z.x, z.y);
            }
        // ---- pass 1 ----
        {
            constexpr int Mq = C::Mp(1), Sin = C::Mp(0) + C::pad(1);
            wave_sync();
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int tau = lane + 64 * s, Kin = tau / Mq, rr = tau - Kin * Mq;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float2 t = lds[rr + Mq * j + Sin * Kin];
                    v[s][j] = cf{t.x, t.y};
                }
            }
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                radix16(v[s]);
                cf w[16], w1 = wp1;
                asm volatile("" : "+v"(w1.x), "+v"(w1.y));
                pow_tree(w1, w);
#pragma unroll
                for (int k = 1; k < 16; ++k) v[s][rev16(k)] = cmul(v[s][rev16(k)], w[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
            constexpr int Sout = Mq + C::pad(2), kstep = C::N / C::Np(1);
            wave_sync();   // all four sets have read the layout entering pass 1
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int tau = lane + 64 * s, Kin = tau / Mq, rr = tau - Kin * Mq;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const cf z = v[s][rev16(k)];
                    lds[rr + Sout * (Kin + kstep * k)] = make_float2(z.x, z.y);
                }
            }
        }
        // ---- pass 2 ----
        {
            constexpr int Mq = C::Mp(2), Sin = C::Mp(1) + C::pad(2);
            wave_sync();
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int tau = lane + 64 * s, Kin = tau / Mq, rr = tau - Kin * Mq;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float2 t = lds[rr + Mq * j + Sin * Kin];
                    v[s][j] = cf{t.x, t.y};
                }
            }
#pragma unroll
            for (int s = 0; s < NV; ++s) radix16(v[s]);
        }
        // ---- epilogue + store: X[tau + 256 q] in v[s][rev16(q)] -> index tau + 256 (q ^ xor) ----
        __amdgpu_buffer_rsrc_t w = frame_rsrc(static_cast<char*>(out_raw) + f * (size_t)(4096 * OUT_ELEM), 4096 * OUT_ELEM);
#pragma unroll
        for (int s = 0; s < NV; ++s)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf z = v[s][rev16(q)];
                if (EPILOGUE == EPI_LOGPSD) {
                    const float db = logpsd_db(z.x, z.y, eps);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, db), w, (lane + 64 * s) * 4, (q ^ xor_q) * 1024, 2);
                } else {
                    v2f o = {z.x, z.y};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), w, (lane + 64 * s) * 8, (q ^ xor_q) * 2048, 0);
                }
            }
    }
}

size_t fft4096_wave_lds_bytes(bool window) { return (size_t)4 * LdsCfg<12>::SLOT * sizeof(float2) + (window ? 4096 * sizeof(float) : 0); }

hipError_t launch_fft4096_wave(const LaunchArgs& a) {
    if (a.n_frames == 0) return hipSuccess;
    const size_t lds_bytes = fft4096_wave_lds_bytes(a.d_window != nullptr);
    size_t groups = (a.n_frames + 3) / 4, cap = (size_t)a.num_cus;
    dim3 g((unsigned)(groups < cap ? groups : cap)), b(256);
    const float2* iq = static_cast<const float2*>(a.d_iq);
    const float2* tw = static_cast<const float2*>(a.d_twiddle);
#define SDRK_LAUNCH(W, E)                                                                                                   \
    do {                                                                                                                    \
        auto kern = fft4096_wave_kernel<W, E>;                                                                              \
        static std::atomic<uint64_t> lds_ok{0};                                                                             \
        hipError_t e0 = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok);                         \
        if (e0 != hipSuccess) return e0;                                                                                    \
        hipLaunchKernelGGL(kern, g, b, lds_bytes, a.stream, iq, a.frame_stride, a.d_out, a.n_frames, a.d_window, tw, a.eps, \
                           a.shift);                                                                                        \
    } while (0)
    if (a.epilogue == EPI_LOGPSD) {
        if (a.d_window) SDRK_LAUNCH(true, EPI_LOGPSD); else SDRK_LAUNCH(false, EPI_LOGPSD);
    } else {
        if (a.d_window) SDRK_LAUNCH(true, EPI_COMPLEX); else SDRK_LAUNCH(false, EPI_COMPLEX);
    }
#undef SDRK_LAUNCH
    return hipGetLastError();
}

}  // namespace sdrk
