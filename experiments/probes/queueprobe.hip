// queueprobe.hip — developer probe: does the 9.2 / 9.8 ms state of the N = 4096 traffic shape depend on the HIP
// stream (hardware queue) a kernel is launched on?  Runs the no-arithmetic 2:1 streaming kernel on 12 streams of one
// process, several rounds, and prints the median per stream.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void mix_k(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_frames) {
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const v4f* x = in + f * 2048;
        v4f* o = out + f * 1024;
        v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[2 * j] + v[2 * j + 1], &o[threadIdx.x + 256 * j]);
    }
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 19;
    const size_t nf = (size_t)1 << lg;
    void *d_in, *d_out;
    CK(hipMalloc(&d_in, nf * 32768)); CK(hipMalloc(&d_out, nf * 16384));
    CK(hipMemset(d_in, 1, nf * 32768)); CK(hipMemset(d_out, 0, nf * 16384));
    const int NS = 12;
    hipStream_t s[NS];
    for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(NS);
    for (int round = 0; round < 6; ++round)
        for (int i = 0; i < NS; ++i) {
            CK(hipEventRecord(e0, s[i]));
            hipLaunchKernelGGL(mix_k, dim3(768), dim3(256), 0, s[i], (const v4f*)d_in, (v4f*)d_out, nf);
            CK(hipEventRecord(e1, s[i]));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (round) ms[i].push_back(t);
        }
    printf("frames=2^%d:", lg);
    for (int i = 0; i < NS; ++i) { std::sort(ms[i].begin(), ms[i].end()); printf(" s%d=%.3f", i, ms[i][ms[i].size() / 2]); }
    printf(" ms  (GB/s on s0: %.0f)\n", 12.0 * nf * 4096 / ms[0][ms[0].size() / 2] / 1e6);
    // second question: separate allocations made later in the same process (other physical pages)?
    const int NB = argc > 2 ? atoi(argv[2]) : 0;
    for (int b = 0; b < NB; ++b) {
        void *pi, *po, *pad;
        CK(hipMalloc(&pad, ((size_t)b * 3 + 1) << 28));          // shift what the allocator hands out next
        CK(hipMalloc(&pi, nf * 32768)); CK(hipMalloc(&po, nf * 16384));
        CK(hipMemset(pi, 1, nf * 32768)); CK(hipMemset(po, 0, nf * 16384));
        std::vector<float> t5;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, s[0]));
            hipLaunchKernelGGL(mix_k, dim3(768), dim3(256), 0, s[0], (const v4f*)pi, (v4f*)po, nf);
            CK(hipEventRecord(e1, s[0]));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r) t5.push_back(t);
        }
        std::sort(t5.begin(), t5.end());
        printf("   alloc %d (in %p out %p): %.3f ms\n", b, pi, po, t5[2]);
    }
    return 0;
}
