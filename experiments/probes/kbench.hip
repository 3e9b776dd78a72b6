// kbench.hip — kernel tuning harness (developer tool, not part of libsdrk.so).
// Compiles the flagship kernel with whatever -DF4K_* switches are given and times
// it with HIP events on device-generated IQ, next to two plain streaming kernels
// with the same bytes (the practical ceiling for this traffic shape).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DF4K_...] kbench.hip -o kbench_<tag>
//   ./kbench_<tag> [log2_frames=20] [reps=20] [window=1]
#include "../../sdr-iq-visualizer_amd/csrc/fft4096.hip"
#include "../../sdr-iq-visualizer_amd/csrc/aux_kernels.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// 8 B/lane loads, 4 B/lane stores, same bytes per frame as the FFT kernel.
__global__ __launch_bounds__(256) void stream_8in_4out(const float2* __restrict__ in, float* __restrict__ out, size_t n_frames) {
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const sdrk::v2f* x = reinterpret_cast<const sdrk::v2f*>(in) + f * 4096;
        float* o = out + f * 4096;
        sdrk::v2f v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]);
#pragma unroll
        for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j].x + v[j].y, &o[threadIdx.x + 256 * j]);
    }
}
// 16 B/lane loads and stores (float4 in -> float2 pairs out packed as float4 per two loads).
__global__ __launch_bounds__(256) void stream_16in_16out(const float4* __restrict__ in, float4* __restrict__ out, size_t n_frames) {
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const sdrk::v4f* x = reinterpret_cast<const sdrk::v4f*>(in) + f * 2048;
        sdrk::v4f* o = reinterpret_cast<sdrk::v4f*>(out) + f * 1024;
        sdrk::v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sdrk::v4f r = v[2 * j] + v[2 * j + 1];
            __builtin_nontemporal_store(r, &o[threadIdx.x + 256 * j]);
        }
    }
}

template <class F>
static void time_it(const char* name, int reps, double bytes, hipStream_t s, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(s));
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    printf("%-28s min %8.3f ms  med %8.3f ms   %7.1f GB/s (med)  %5.1f %% of 8 TB/s\n", name, ms[0], ms[reps / 2],
           bytes / ms[reps / 2] / 1e6, bytes / ms[reps / 2] / 1e6 / 80.0);
}

int main(int argc, char** argv) {
    int lg = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 20, win = argc > 3 ? atoi(argv[3]) : 1;
    size_t nf = (size_t)1 << lg;
    void *d_in, *d_out; float* d_win; float2* d_tw;
    CK(hipMalloc(&d_in, nf * 4096 * 8)); CK(hipMalloc(&d_out, nf * 4096 * 4));
    CK(hipMalloc(&d_win, 4096 * 4)); CK(hipMalloc(&d_tw, 4096 * 8));
    std::vector<float> w(4096); std::vector<float2> tw(4096);
    for (int n = 0; n < 4096; ++n) { w[n] = (float)(0.5 - 0.5 * cos(2 * M_PI * n / 4095.0)); double a = -2 * M_PI * n / 4096.0; tw[n] = make_float2((float)cos(a), (float)sin(a)); }
    CK(hipMemcpy(d_win, w.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tw, tw.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(sdrk::launch_synth_fill(1234, 0, nf, 4096, d_in, s)); CK(hipStreamSynchronize(s));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    double bytes = 12.0 * nf * 4096;
    printf("d_in=%p d_out=%p\n", d_in, d_out);
    printf("variant: WAVES=%d NT=%d  frames=2^%d window=%d\n", F4K_WAVES, F4K_NT, lg, win);
    sdrk::LaunchArgs a; a.d_iq = d_in; a.frame_stride = 4096; a.d_out = d_out; a.n_frames = nf; a.nfft = 4096;
    a.d_window = win ? d_win : nullptr; a.d_twiddle = d_tw; a.stream = s; a.num_cus = prop.multiProcessorCount;
    for (int round = 0; round < 3; ++round) {   // interleaved A/B on the same buffers
        time_it("fft4096", reps, bytes, s, [&] { CK(sdrk::launch_fft4096(a)); });
        time_it("no-arithmetic 2:1 stream (same buffers)", reps, bytes, s, [&] { CK(sdrk::launch_stream_mix(d_in, d_out, nf, prop.multiProcessorCount, s)); });
    }
    if (argc > 5) {   // single-slab experiment: input and output carved from ONE allocation at chosen offsets
        void* slab; const size_t in_b = nf * 4096 * 8, out_b = nf * 4096 * 4;
        CK(hipMalloc(&slab, in_b + out_b + ((size_t)1 << 30)));
        CK(hipMemcpyAsync(slab, d_in, in_b, hipMemcpyDeviceToDevice, s)); CK(hipStreamSynchronize(s));
        a.d_iq = slab;
        for (size_t off : {(size_t)0, (size_t)256, (size_t)4096, (size_t)32768, (size_t)65536, (size_t)1 << 20, (size_t)2 << 20,
                           (size_t)3 << 20, (size_t)16 << 20, (size_t)100 << 20, (size_t)512 << 20}) {
            a.d_out = (char*)slab + in_b + off;
            char name[64]; snprintf(name, sizeof name, "  slab: out at in_end +%zu KiB", off >> 10);
            time_it(name, reps, bytes, s, [&] { CK(sdrk::launch_fft4096(a)); });
        }
        a.d_iq = d_in; a.d_out = d_out;
        CK(hipFree(slab));
    } else if (argc > 4) {   // placement experiment: same kernel, output (or input) buffer shifted / re-allocated
        for (size_t off : {(size_t)4096, (size_t)65536, (size_t)1 << 20, (size_t)3 << 20, (size_t)64 << 20}) {
            void* d_out2; CK(hipMalloc(&d_out2, nf * 4096 * 4 + off + (128 << 20)));
            a.d_out = (char*)d_out2 + off;
            char name[64]; snprintf(name, sizeof name, "  new out alloc +%zu KiB", off >> 10);
            time_it(name, reps, bytes, s, [&] { CK(sdrk::launch_fft4096(a)); });
            CK(hipFree(d_out2));
        }
        a.d_out = d_out;
        time_it("  original buffers again", reps, bytes, s, [&] { CK(sdrk::launch_fft4096(a)); });
    }
    // checksum of a few outputs so variants can be compared for equality
    std::vector<float> h(4096 * 2);
    CK(hipMemcpy(h.data(), (float*)d_out + (nf - 2) * 4096, 4096 * 2 * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : h) cs += v;
    printf("checksum(last two rows) = %.6f   row[-1][0..3] = %.5f %.5f %.5f %.5f\n", cs, h[4096], h[4097], h[4098], h[4099]);
    unsigned grid = prop.multiProcessorCount * 8;
    time_it("stream 8B in / 4B out", reps, bytes, s, [&] { hipLaunchKernelGGL(stream_8in_4out, dim3(grid), dim3(256), 0, s, (const float2*)d_in, (float*)d_out, nf); });
    time_it("stream 16B in / 16B out", reps, bytes, s, [&] { hipLaunchKernelGGL(stream_16in_16out, dim3(grid), dim3(256), 0, s, (const float4*)d_in, (float4*)d_out, nf); });
    return 0;
}
