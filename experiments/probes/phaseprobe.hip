// phaseprobe.hip — does separating reads and writes in TIME, chip-wide, buy HBM bandwidth for a 2:1 stream (32 KiB read, 16 KiB
// written per frame, the flagship's traffic shape without its arithmetic)?  Every workgroup gates the ISSUE of its loads to a "read
// window" and of its stores to a "write window" of a period P, told by the constant-rate real-time counter all CUs share
// (s_memrealtime, 100 MHz) — no communication.  Variant 0: no gating (the software-pipelined copy).  Developer tool:
//   hipcc --offload-arch=gfx950 -O3 -o phaseprobe phaseprobe.hip && ./phaseprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// wait until (now mod period) lies in [lo, hi)
__device__ __forceinline__ void wait_window(unsigned period, unsigned lo, unsigned hi) {
    for (int guard = 0; guard < 2000; ++guard) {            // bounded (~0.1 ms): a window always comes within one period
        const unsigned t = (unsigned)(__builtin_amdgcn_s_memrealtime() % period);
        if (t >= lo && t < hi) return;
        __builtin_amdgcn_s_sleep(2);
    }
}

template <bool PHASED>
__global__ __launch_bounds__(256, 3) void copy21(const float2* __restrict__ in, float* __restrict__ out, size_t n_frames,
                                                 unsigned period, unsigned read_end) {
    const int tid = threadIdx.x;
    const size_t first = blockIdx.x, step = gridDim.x;
    v2u nxt[16];
    auto issue = [&](size_t fr) {
        if (fr >= n_frames) fr = first;
        __amdgpu_buffer_rsrc_t r = rsrc(in + fr * 4096, 4096 * 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) nxt[j] = __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, j * 2048, 2);
    };
    if (PHASED) wait_window(period, 0, read_end);
    issue(first);
    for (size_t f = first; f < n_frames; f += step) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_bit_cast(float, nxt[j].x) + __builtin_bit_cast(float, nxt[j].y);
        if (PHASED) wait_window(period, 0, read_end);
        issue(f + step);
        if (PHASED) wait_window(period, read_end, period);
        __amdgpu_buffer_rsrc_t w = rsrc(out + f * 4096, 4096 * 4);
#pragma unroll
        for (int j = 0; j < 16; ++j) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), w, tid * 4, j * 1024, 2);
    }
}

// the ungated copy with other access widths: LW = bytes per lane and load (8 / 16), SW = bytes per lane and store (4 / 8 / 16)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int LW, int SW>
__global__ __launch_bounds__(256, 3) void copy21w(const float2* __restrict__ in, float* __restrict__ out, size_t n_frames) {
    const int tid = threadIdx.x;
    const size_t first = blockIdx.x, step = gridDim.x;
    unsigned nxt[32];
    auto issue = [&](size_t fr) {
        if (fr >= n_frames) fr = first;
        __amdgpu_buffer_rsrc_t r = rsrc(in + fr * 4096, 4096 * 8);
        if (LW == 8) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { v2u t = __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, j * 2048, 2); nxt[2 * j] = t.x; nxt[2 * j + 1] = t.y; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, j * 4096, 2); nxt[4 * j] = t.x; nxt[4 * j + 1] = t.y; nxt[4 * j + 2] = t.z; nxt[4 * j + 3] = t.w; }
        }
    };
    issue(first);
    for (size_t f = first; f < n_frames; f += step) {
        unsigned v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = nxt[2 * j] ^ nxt[2 * j + 1];
        issue(f + step);
        __amdgpu_buffer_rsrc_t w = rsrc(out + f * 4096, 4096 * 4);
        if (SW == 4) {
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_amdgcn_raw_buffer_store_b32(v[j], w, tid * 4, j * 1024, 2);
        } else if (SW == 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b64(v2u{v[2 * j], v[2 * j + 1]}, w, tid * 8, j * 2048, 2);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(v4u{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]}, w, tid * 16, j * 4096, 2);
        }
    }
}

// the workgroup-per-frame copy with DEPTH frames prefetched (registers) and PER_CU workgroups per CU
template <int DEPTH, int PER_CU>
__global__ __launch_bounds__(256, PER_CU) void copy21d(const float2* __restrict__ in, float* __restrict__ out, size_t n_frames) {
    const int tid = threadIdx.x;
    const size_t first = blockIdx.x, step = gridDim.x;
    v2u q[DEPTH][16];
    auto issue = [&](v2u (&x)[16], size_t fr) {
        if (fr >= n_frames) fr = first;
        __amdgpu_buffer_rsrc_t r = rsrc(in + fr * 4096, 4096 * 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8, j * 2048, 2);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(q[d], first + d * step);
    for (size_t f = first; f < n_frames; f += step * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t fd = f + d * step;
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_bit_cast(float, q[d][j].x) + __builtin_bit_cast(float, q[d][j].y);
            issue(q[d], fd + step * DEPTH);
            if (fd < n_frames) {
                __amdgpu_buffer_rsrc_t w = rsrc(out + fd * 4096, 4096 * 4);
#pragma unroll
                for (int j = 0; j < 16; ++j) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), w, tid * 4, j * 1024, 2);
            }
        }
    }
}

// the same traffic with ONE WAVE per frame (64 points per lane, prefetched one frame ahead: 32 KiB in flight per wave), WPC waves
// per CU: the memory side of a wave-per-frame transform kernel
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void copy21_wave(const float2* __restrict__ in, float* __restrict__ out, size_t n_frames) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t first = (size_t)blockIdx.x * 4 + wave, step = (size_t)gridDim.x * 4;
    v2u nxt[64];
    auto issue = [&](size_t fr) {
        if (fr >= n_frames) fr = first;
        __amdgpu_buffer_rsrc_t r = rsrc(in + fr * 4096, 4096 * 8);
#pragma unroll
        for (int j = 0; j < 64; ++j) nxt[j] = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, j * 512, 2);
    };
    issue(first);
    for (size_t f = first; f < n_frames; f += step) {
        float v[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) v[j] = __builtin_bit_cast(float, nxt[j].x) + __builtin_bit_cast(float, nxt[j].y);
        issue(f + step);
        __amdgpu_buffer_rsrc_t w = rsrc(out + f * 4096, 4096 * 4);
#pragma unroll
        for (int j = 0; j < 64; ++j) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), w, lane * 4, j * 256, 2);
    }
}

int main(int argc, char** argv) {
    const size_t frames = (size_t)1 << 19;                   // 16 GiB in, 8 GiB out
    float2* in; float* out;
    if (hipMalloc(&in, frames * 4096 * 8) != hipSuccess || hipMalloc(&out, frames * 4096 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 0, frames * 4096 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](int phased, unsigned period, unsigned read_end) {
        auto launch = [&] {
            if (phased) hipLaunchKernelGGL(copy21<true>, dim3(768), dim3(256), 0, 0, in, out, frames, period, read_end);
            else hipLaunchKernelGGL(copy21<false>, dim3(768), dim3(256), 0, 0, in, out, frames, period, read_end);
        };
        for (int i = 0; i < 12; ++i) launch();               // > 50 ms of load: the shader clock has ramped
        (void)hipDeviceSynchronize();
        float best = 1e30f, sum = 0;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
        }
        const double gb = (double)frames * 4096 * 12 / 1e9;
        if (phased) printf("phased  period %5.2f us  read window %4.0f %%   ", period / 100.0, 100.0 * read_end / period);
        else printf("plain (software-pipelined, no gating)              ");
        printf("%7.3f ms best %7.3f mean   %6.0f GB/s\n", best, sum / 5, gb / best * 1e3);
        fflush(stdout);
    };
    if (argc > 1) {      // access widths of the ungated copy
        auto runw = [&](const char* name, void (*k)(const float2*, float*, size_t)) {
            for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(k, dim3(768), dim3(256), 0, 0, in, out, frames);
            (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 7; ++r) {
                (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(768), dim3(256), 0, 0, in, out, frames); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%-40s %7.3f ms   %6.0f GB/s\n", name, best, (double)frames * 4096 * 12 / 1e9 / best * 1e3); fflush(stdout);
        };
        if (argv[1][0] == 'v') {     // "vave": one wave per frame against the workgroup-per-frame copy, alternating
            auto runv = [&](const char* name, int grid) {
                for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(copy21_wave<0>, dim3(grid), dim3(256), 0, 0, in, out, frames);
                (void)hipDeviceSynchronize();
                float best = 1e30f;
                for (int r = 0; r < 7; ++r) {
                    (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(copy21_wave<0>, dim3(grid), dim3(256), 0, 0, in, out, frames); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("%-40s %7.3f ms   %6.0f GB/s\n", name, best, (double)frames * 4096 * 12 / 1e9 / best * 1e3); fflush(stdout);
            };
            auto rund = [&](const char* name, void (*k)(const float2*, float*, size_t), int grid) {
                for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, frames);
                (void)hipDeviceSynchronize();
                float best = 1e30f;
                for (int r = 0; r < 7; ++r) {
                    (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, frames); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("%-40s %7.3f ms   %6.0f GB/s\n", name, best, (double)frames * 4096 * 12 / 1e9 / best * 1e3); fflush(stdout);
            };
            for (int rep = 0; rep < 2; ++rep) {
                rund("workgroup per frame: 3 per CU, depth 1", copy21d<1, 3>, 768);
                rund("workgroup per frame: 2 per CU, depth 1", copy21d<1, 2>, 512);
                rund("workgroup per frame: 2 per CU, depth 2", copy21d<2, 2>, 512);
                rund("workgroup per frame: 3 per CU, depth 2", copy21d<2, 3>, 768);
                rund("workgroup per frame: 1 per CU, depth 4", copy21d<4, 1>, 256);
                rund("workgroup per frame: 1 per CU, depth 2", copy21d<2, 1>, 256);
                rund("workgroup per frame: 4 per CU, depth 1", copy21d<1, 4>, 1024);
            }
            for (int rep = 0; rep < 2; ++rep) {
                runw("workgroup per frame, 3 per CU (ships)", copy21w<8, 4>);
                runv("wave per frame, 4 waves per CU", 256);
                runv("wave per frame, 4 waves per CU, grid x2", 512);
            }
            return 0;
        }
        for (int rep = 0; rep < 2; ++rep) {
            runw("loads  8 B/lane, stores  4 B/lane", copy21w<8, 4>);
            runw("loads 16 B/lane, stores  4 B/lane", copy21w<16, 4>);
            runw("loads  8 B/lane, stores  8 B/lane", copy21w<8, 8>);
            runw("loads 16 B/lane, stores  8 B/lane", copy21w<16, 8>);
            runw("loads 16 B/lane, stores 16 B/lane", copy21w<16, 16>);
            runw("loads  8 B/lane, stores 16 B/lane", copy21w<8, 16>);
        }
        return 0;
    }
    run(0, 1, 1);
    for (unsigned period : {300u, 450u, 600u, 800u, 1200u, 2000u})
        for (unsigned pct : {50u, 60u, 67u})
            run(1, period, period * pct / 100);
    run(0, 1, 1);
    return 0;
}
