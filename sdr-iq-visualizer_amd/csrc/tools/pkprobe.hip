// pkprobe.hip — issue rate of the packed float32 VALU forms a complex butterfly would use (v_pk_add_f32 with and without
// the half-swap / negate modifiers of a multiplication by -i, v_pk_mul_f32, v_pk_fma_f32) against v_add_f32 / v_fma_f32,
// at 1, 2 and 3 waves per SIMD.  Developer tool: hipcc --offload-arch=gfx950 -O3 -o pkprobe pkprobe.hip && ./pkprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    v2f a[16], c = {seed, seed * 0.5f}, d = {1.0f - seed, seed};
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = v2f{seed * k, seed + k};
    for (int it = 0; it < iters; ++it) {
#define ONE(k)                                                                                                              \
    if (V == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k].x) : "v"(c.x), "v"(d.x)); }                          \
    if (V == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(d)); }                             \
    if (V == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }                                         \
    if (V == 3) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[k]) : "v"(c)); } \
    if (V == 4) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }                                         \
    if (V == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(c.x)); }                                        \
    if (V == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(a[k]) : "v"(c), "v"(d)); } \
    if (V == 7) { asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(a[k].x), "+v"(a[k].y) : "v"(c.x), "v"(c.y)); }
            REP16(ONE) REP16(ONE) REP16(ONE) REP16(ONE)
#undef ONE
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k].x + a[k].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int V>
static void run(const char* name, int per_iter_mult, float* d_out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg = 1; wg <= 3; ++wg) {
        dim3 g(256 * wg), b(256);
        // warm up by time: the shader clock needs tens of milliseconds of load
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(probe<V>, g, b, 0, 0, d_out, iters, 0.001f);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe<V>, g, b, 0, 0, d_out, iters, 0.001f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // instructions one SIMD issued: wg waves (one wave of each workgroup per SIMD) * iters * 64 (* 2 for the two-add form)
        const double instr = (double)wg * iters * 64 * per_iter_mult;
        printf("%-44s waves/SIMD %d  %8.3f ms  %6.2f ns per instruction per SIMD (= cycles at 1 GHz; x sclk GHz = cycles)\n", name, wg,
               best, best * 1e6 / instr);
    }
}

int main() {
    float* d_out; hipMalloc(&d_out, 4096);
    run<5>("v_add_f32", 1, d_out);
    run<0>("v_fma_f32", 1, d_out);
    run<7>("2 x v_add_f32 (one complex add)", 2, d_out);
    run<2>("v_pk_add_f32", 1, d_out);
    run<3>("v_pk_add_f32 op_sel swap + neg_hi (a - i b)", 1, d_out);
    run<4>("v_pk_mul_f32", 1, d_out);
    run<1>("v_pk_fma_f32", 1, d_out);
    run<6>("v_pk_fma_f32 op_sel + neg (complex mul half)", 1, d_out);
    return 0;
}
