// streambench.hip — what can a plain streaming kernel reach on this box with the
// FFT path's traffic shape (8 B/sample read, 4 B/sample written)?  Developer tool.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// frame = 4096 complex64 (32 KiB) in, 4096 float (16 KiB) out; one 256-thread WG per frame.
// MODE 0: frames interleaved over blocks (f = b, b+G, ...); MODE 1: each block owns a contiguous chunk.
template <int NT, int MODE, int DO_READ, int DO_WRITE>
__global__ __launch_bounds__(256) void stream_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_frames) {
    size_t f0, f1, step;
    if (MODE == 0) { f0 = blockIdx.x; f1 = n_frames; step = gridDim.x; }
    else { size_t per = (n_frames + gridDim.x - 1) / gridDim.x; f0 = per * blockIdx.x; f1 = f0 + per < n_frames ? f0 + per : n_frames; step = 1; }
    v4f acc = {0, 0, 0, 0};
    for (size_t f = f0; f < f1; f += step) {
        const v4f* x = in + f * 2048;
        v4f* o = out + f * 1024;
        v4f v[8];
        if (DO_READ) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]) : x[threadIdx.x + 256 * j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v4f{(float)f, 1, 2, 3};
        }
        if (DO_WRITE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v4f r = v[2 * j] + v[2 * j + 1]; if (NT) __builtin_nontemporal_store(r, &o[threadIdx.x + 256 * j]); else o[threadIdx.x + 256 * j] = r; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    if (!DO_WRITE && acc.x == 12345.678f) out[0] = acc;
}

// TDMA experiment: the whole chip reads during [0,TR) and writes during [TR,TR+TW) of a
// period measured on the shared 100 MHz s_memrealtime clock, so that HBM sees long
// read-only and write-only bursts instead of a fine-grained mix.
__global__ __launch_bounds__(256) void tdma_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_frames,
                                              unsigned TR, unsigned TW) {
    const unsigned period = TR + TW;
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const v4f* x = in + f * 2048;
        v4f* o = out + f * 1024;
        v4f v[8];
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() % period) >= TR) __builtin_amdgcn_s_sleep(4);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]);
        v4f r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = v[2 * j] + v[2 * j + 1];   // forces the loads to complete
        asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() % period) < TR) __builtin_amdgcn_s_sleep(4);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(r[j], &o[threadIdx.x + 256 * j]);
    }
}

// cache-policy sweep with buffer instructions: LD_AUX / ST_AUX = sc0 (1) | nt (2) | sc1 (16)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int LD_AUX, int ST_AUX>
__global__ __launch_bounds__(256) void policy_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_frames) {
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(in + f * 2048), 0, 32768, 0x00020000);
        __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)(out + f * 1024), 0, 16384, 0x00020000);
        v4u v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, j * 4096, LD_AUX);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[2 * j] ^ v[2 * j + 1], w, threadIdx.x * 16, j * 4096, ST_AUX);
    }
}

// classic copy: n float4 in -> n float4 out, grid-stride
template <int NT>
__global__ __launch_bounds__(256) void copy_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4f v = NT ? __builtin_nontemporal_load(&in[i]) : in[i];
        if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    }
}

template <class F>
static float time_it(int reps, hipStream_t s, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipStreamSynchronize(s));
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[i], e0, e1)); }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}

int main(int argc, char** argv) {
    int lg = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 10;
    size_t nf = (size_t)1 << lg;
    void *d_in, *d_out;
    CK(hipMalloc(&d_in, nf * 4096 * 8)); CK(hipMalloc(&d_out, nf * 4096 * 8));   // out big enough for the 1:1 copy too
    CK(hipMemset(d_in, 1, nf * 4096 * 8)); CK(hipMemset(d_out, 0, nf * 4096 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("frames=2^%d (%.1f GiB in, %.1f GiB out)\n", lg, nf * 32768.0 / (1 << 30), nf * 16384.0 / (1 << 30));
    const v4f* in = (const v4f*)d_in; v4f* out = (v4f*)d_out;
    for (int bpc : {2, 3, 4, 6, 8}) {
        unsigned g = cus * bpc;
        double rw = 12.0 * nf * 4096, ro = 8.0 * nf * 4096, wo = 4.0 * nf * 4096;
        float t;
        t = time_it(reps, s, [&] { hipLaunchKernelGGL((stream_k<1, 0, 1, 1>), dim3(g), dim3(256), 0, s, in, out, nf); });
        printf("bpc %d  r+w nt interleaved   %8.3f ms %7.1f GB/s\n", bpc, t, rw / t / 1e6);
        t = time_it(reps, s, [&] { hipLaunchKernelGGL((stream_k<0, 0, 1, 1>), dim3(g), dim3(256), 0, s, in, out, nf); });
        printf("bpc %d  r+w plain interleaved %8.3f ms %7.1f GB/s\n", bpc, t, rw / t / 1e6);
        t = time_it(reps, s, [&] { hipLaunchKernelGGL((stream_k<1, 1, 1, 1>), dim3(g), dim3(256), 0, s, in, out, nf); });
        printf("bpc %d  r+w nt chunked        %8.3f ms %7.1f GB/s\n", bpc, t, rw / t / 1e6);
        t = time_it(reps, s, [&] { hipLaunchKernelGGL((stream_k<1, 0, 1, 0>), dim3(g), dim3(256), 0, s, in, out, nf); });
        printf("bpc %d  read-only nt          %8.3f ms %7.1f GB/s\n", bpc, t, ro / t / 1e6);
        t = time_it(reps, s, [&] { hipLaunchKernelGGL((stream_k<1, 0, 0, 1>), dim3(g), dim3(256), 0, s, in, out, nf); });
        printf("bpc %d  write-only nt         %8.3f ms %7.1f GB/s\n", bpc, t, wo / t / 1e6);
    }
    for (int bpc : {3, 4, 6, 8}) {
        unsigned g = cus * bpc;
        double rw = 12.0 * nf * 4096;
        for (auto tr_tw : {std::pair<unsigned, unsigned>{175, 110}, {350, 220}, {700, 440}, {1400, 880}, {350, 350}, {240, 240}}) {
            float t = time_it(reps, s, [&] { hipLaunchKernelGGL(tdma_k, dim3(g), dim3(256), 0, s, in, out, nf, tr_tw.first, tr_tw.second); });
            printf("bpc %d  TDMA TR=%4u TW=%4u (x10 ns)   %8.3f ms %7.1f GB/s\n", bpc, tr_tw.first, tr_tw.second, t, rw / t / 1e6);
        }
    }
    {
        unsigned g = cus * 3;
        double rw = 12.0 * nf * 4096;
        float t;
#define POL(L, S) t = time_it(reps, s, [&] { hipLaunchKernelGGL((policy_k<L, S>), dim3(g), dim3(256), 0, s, in, out, nf); }); \
        printf("bpc 3 policy ld_aux=%2d st_aux=%2d   %8.3f ms %7.1f GB/s\n", L, S, t, rw / t / 1e6);
        for (int rep = 0; rep < 2; ++rep) {
            POL(2, 2) POL(0, 0) POL(2, 0) POL(0, 2) POL(2, 16) POL(2, 18) POL(2, 17) POL(2, 19) POL(16, 2) POL(18, 2) POL(1, 2) POL(2, 1) POL(2, 3)
        }
    }
    size_t n4 = nf * 2048;  // float4 count of the input
    for (int bpc : {4, 8, 16}) {
        unsigned g = cus * bpc;
        float t = time_it(reps, s, [&] { hipLaunchKernelGGL((copy_k<0>), dim3(g), dim3(256), 0, s, in, out, n4); });
        printf("copy float4 plain bpc %2d   %8.3f ms %7.1f GB/s (r+w)\n", bpc, t, 2.0 * n4 * 16 / t / 1e6);
        t = time_it(reps, s, [&] { hipLaunchKernelGGL((copy_k<1>), dim3(g), dim3(256), 0, s, in, out, n4); });
        printf("copy float4 nt    bpc %2d   %8.3f ms %7.1f GB/s (r+w)\n", bpc, t, 2.0 * n4 * 16 / t / 1e6);
    }
    return 0;
}
