// orderprobe.hip — developer probe: does the ORDER in which the persistent workgroups walk the frames change the
// streaming rate?  grid-stride (frame = b, b + G, ...: one contiguous moving window of G frames) against blocked
// runs (workgroup b owns frames [b R, (b+1) R): G windows spread over the whole buffer) against XCD-blocked
// (the 8 XCDs own eighths of the buffer, grid-stride inside).  32 KiB read + 16 KiB written per frame, and
// write-only, over K allocation pairs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int RD, int ORDER>
__global__ __launch_bounds__(256) void mix_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_frames) {
    const size_t G = gridDim.x, b = blockIdx.x;
    size_t f0, f1, step;
    if (ORDER == 0) { f0 = b; f1 = n_frames; step = G; }
    else if (ORDER == 1) { const size_t R = (n_frames + G - 1) / G; f0 = b * R; f1 = f0 + R < n_frames ? f0 + R : n_frames; step = 1; }
    else { const size_t E = n_frames / 8, x = b & 7, j = b >> 3; f0 = x * E + j; f1 = (x + 1) * E; step = G / 8; }
    for (size_t f = f0; f < f1; f += step) {
        const v4f* x = in + f * 2048;
        v4f* o = out + f * 1024;
        v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = RD ? __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]) : v4f{(float)f, 1, 2, 3};
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[2 * j] + v[2 * j + 1], &o[threadIdx.x + 256 * j]);
    }
}
static hipEvent_t e0, e1;
template <int RD, int ORDER>
static float timeit(const void* in, void* out, size_t nf) {
    std::vector<float> t;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mix_k<RD, ORDER>), dim3(768), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, nf);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20, K = argc > 2 ? atoi(argv[2]) : 4;
    const size_t nf = (size_t)1 << lg;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < K; ++i) {
        void *in, *out;
        CK(hipMalloc(&in, nf * 32768)); CK(hipMalloc(&out, nf * 16384));
        CK(hipMemset(in, 1, nf * 32768)); CK(hipMemset(out, 0, nf * 16384));
        printf("pair %d  read+write ms: stride %.3f  blocked %.3f  xcd-blocked %.3f   | write-only: stride %.3f  blocked %.3f  xcd-blocked %.3f\n", i,
               timeit<1, 0>(in, out, nf), timeit<1, 1>(in, out, nf), timeit<1, 2>(in, out, nf),
               timeit<0, 0>(in, out, nf), timeit<0, 1>(in, out, nf), timeit<0, 2>(in, out, nf));
        fflush(stdout);
        // keep the pair allocated so that the next one lands elsewhere
    }
    return 0;
}
