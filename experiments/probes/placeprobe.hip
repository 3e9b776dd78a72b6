// placeprobe.hip — developer probe: the 6 % two-state behaviour of the N = 4096 traffic shape follows the BUFFERS
// (queueprobe.hip: not the stream, not the process as such).  Which buffer — input, output, or the pair?
// K inputs x K outputs, each pair timed with the no-arithmetic 2:1 streaming kernel; read-only and write-only too.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int RD, int WR>
__global__ __launch_bounds__(256) void mix_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_frames) {
    v4f acc = {0, 0, 0, 0};
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const v4f* x = in + f * 2048;
        v4f* o = out + f * 1024;
        v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = RD ? __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]) : v4f{(float)f, 1, 2, 3};
        if (WR) {
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[2 * j] + v[2 * j + 1], &o[threadIdx.x + 256 * j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    if (!WR && acc.x == 12345.678f) out[0] = acc;
}
static hipEvent_t e0, e1;
template <int RD, int WR>
static float timeit(const void* in, void* out, size_t nf) {
    std::vector<float> t;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mix_k<RD, WR>), dim3(768), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, nf);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 18, K = argc > 2 ? atoi(argv[2]) : 4;
    const size_t nf = (size_t)1 << lg;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<void*> in(K), out(K);
    for (int i = 0; i < K; ++i) {
        void* pad; CK(hipMalloc(&pad, ((size_t)i * 5 + 1) << 27));
        CK(hipMalloc(&in[i], nf * 32768)); CK(hipMalloc(&out[i], nf * 16384));
        CK(hipMemset(in[i], 1, nf * 32768)); CK(hipMemset(out[i], 0, nf * 16384));
    }
    printf("frames=2^%d  read-only per input:", lg);
    for (int i = 0; i < K; ++i) printf(" %.3f", timeit<1, 0>(in[i], out[0], nf));
    printf("\n              write-only per output:");
    for (int j = 0; j < K; ++j) printf(" %.3f", timeit<0, 1>(in[0], out[j], nf));
    printf("\n              read+write, rows = input, columns = output (ms):\n");
    for (int i = 0; i < K; ++i) {
        printf("   in%d %p:", i, in[i]);
        for (int j = 0; j < K; ++j) printf(" %.3f", timeit<1, 1>(in[i], out[j], nf));
        printf("\n");
    }
    if (argc > 3) {   // does a prefix of the pair predict the whole pair?  (argv[3] = log2 of the prefix frames)
        const size_t pf = (size_t)1 << atoi(argv[3]);
        printf("   prefix of 2^%d frames, same pairs (ms):\n", atoi(argv[3]));
        for (int i = 0; i < K; ++i) {
            printf("   in%d:", i);
            for (int j = 0; j < K; ++j) printf(" %.4f", timeit<1, 1>(in[i], out[j], pf));
            printf("\n");
        }
        printf("   whole pairs again (stability):\n");
        for (int i = 0; i < K; ++i) {
            printf("   in%d:", i);
            for (int j = 0; j < K; ++j) printf(" %.3f", timeit<1, 1>(in[i], out[j], nf));
            printf("\n");
        }
    }
    printf("   outputs:");
    for (int j = 0; j < K; ++j) printf(" %p", out[j]);
    printf("\n");
    return 0;
}
