// VALU issue-rate probe: N dependent-chain-free v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 (with and without op_sel/neg modifiers)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v2f a[8], b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.y), "v"(c.y)); }
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(b));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y)); }
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static int run(const char* name, float* d, int pairs_per_iter) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // "pair-ops" = operations on one (x, y) pair per lane
    const double pairs = (double)grid * 256 * iters * 64.0;
    printf("%-46s %8.3f ms   %.2f T pair-ops/s\n", name, ms, pairs / ms / 1e9);
    return 0;
}
int main() {
    float* d; CK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run<0>("2 x v_fma_f32 per pair", d, 0);
    run<1>("1 x v_pk_fma_f32 per pair", d, 0);
    run<5>("2 x v_add_f32 per pair", d, 0);
    run<2>("1 x v_pk_add_f32 per pair", d, 0);
    run<3>("1 x v_pk_add_f32 with op_sel swap + neg_hi", d, 0);
    run<4>("1 x v_pk_fma_f32 with op_sel + neg_lo", d, 0);
    return 0;
}
