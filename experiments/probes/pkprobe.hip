// pkprobe.hip — issue rate of the packed float32 VALU forms a complex butterfly would use (v_pk_add_f32 with and without
// the half-swap / negate modifiers of a multiplication by -i, v_pk_mul_f32, v_pk_fma_f32) against v_add_f32 / v_fma_f32,
// at 1, 2 and 3 waves per SIMD.  Developer tool: hipcc --offload-arch=gfx950 -O3 -o pkprobe pkprobe.hip && ./pkprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
    if (V == 7) { asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(a[k].x), "+v"(a[k].y) : "v"(c.x), "v"(c.y)); } \
    if (V == 8) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }                                            \
    if (V == 9) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(d)); }                                \
    if (V == 10) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(c.x)); }                                         \
    if (V == 11) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k].x) : "v"(c.x)); }                                       \
    if (V == 12) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k].x) : "v"(c.x) : ); }                           \
    if (V == 13) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[k].x), "v"(c.x) : "vcc"); }                           \
    if (V == 14) { asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k].x)); }             \
    if (V == 15) { asm volatile("v_log_f32 %0, %0" : "+v"(a[k].x)); }                                                      \
    if (V == 16) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(c.x)); }                                       \
    if (V == 17) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k].x)); }                                               \
    if (V == 18) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }                                           \
    if (V == 19) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(c.x)); }                                       \
    if (V == 20) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k].x) : "v"(c.x)); }                                       \
    if (V == 21) { asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[k].x) : "s20"); }                                    \
    if (V == 22) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k].x) : "v"(d)); }                                         \
    if (V == 24) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[k].x) : "v"(c.x) : "s20", "s21"); }           \
    if (V == 25) { asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k].x) : "v"(c.x) : "vcc"); } \
    if (V == 26) { asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[k].x) : "v"(c.x) : "s20", "s21"); } \
    if (V == 27) { asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[k].x) : "v"(c.x) : "vcc"); }                     \
    if (V == 28) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k].x) : "v"(c.x), "v"(d.x)); }                        \
    if (V == 29) { asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_mov_b32 %0, %1\n\ts_mov_b64 exec, s[20:21]" : "+v"(a[k].x) : "v"(c.x) : "vcc", "s20", "s21", "scc"); } \
    if (V == 30) { asm volatile("v_max_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %2" : "+v"(a[k].x) : "v"(c.x), "v"(d.x)); }       \
    if (V == 31) { asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k].x) : "v"(c.x), "v"(d.x)); }                            \
    if (V == 32) { asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k].x) : "v"(c.x), "v"(d.x)); }                             \
    if (V == 23) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[k].x) : "v"(c.x) : "vcc"); }
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
    for (int wg = 1; wg <= 4; ++wg) {
        dim3 g(256 * wg), b(256);
        // warm up by time: the shader clock needs tens of milliseconds of load
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe<V>, g, b, 0, 0, d_out, iters, 0.001f);   // >= 50 ms
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
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    float* d_out; hipMalloc(&d_out, 4096);
    if (argc > 1) {   // one of the mask-reading forms, by number (each in a process of its own, under its own timeout)
        switch (atoi(argv[1])) {
            case 12: run<12>("v_cndmask_b32 (vcc)", 1, d_out); break;
            case 28: run<28>("v_cndmask_b32 (vcc), independent destination", 1, d_out); break;
            case 24: run<24>("v_cndmask_b32_e64 (sgpr pair mask)", 1, d_out); break;
            case 25: run<25>("v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (pair)", 2, d_out); break;
            case 26: run<26>("v_cmp_gt_f32 s[..] + v_cndmask_b32_e64 s[..] (pair)", 2, d_out); break;
            case 27: run<27>("v_addc_co_u32 (vcc in, vcc out)", 1, d_out); break;
            case 29: run<29>("v_cmp + s_and_saveexec + v_mov + s_mov exec (4 instr)", 4, d_out); break;
            case 30: run<30>("v_max_f32 + v_min_f32 (pair)", 2, d_out); break;
            case 31: run<31>("v_med3_f32", 1, d_out); break;
            case 32: run<32>("v_bfi_b32", 1, d_out); break;
        }
        return 0;
    }
    run<5>("v_add_f32", 1, d_out);
    run<0>("v_fma_f32", 1, d_out);
    run<7>("2 x v_add_f32 (one complex add)", 2, d_out);
    run<2>("v_pk_add_f32", 1, d_out);
    run<3>("v_pk_add_f32 op_sel swap + neg_hi (a - i b)", 1, d_out);
    run<4>("v_pk_mul_f32", 1, d_out);
    run<1>("v_pk_fma_f32", 1, d_out);
    run<6>("v_pk_fma_f32 op_sel + neg (complex mul half)", 1, d_out);
    run<19>("v_mul_f32", 1, d_out);
    run<16>("v_max_f32", 1, d_out);
    run<11>("v_add_u32", 1, d_out);
    run<20>("v_and_b32", 1, d_out);
    run<17>("v_lshlrev_b32", 1, d_out);
    run<12>("v_cndmask_b32 (vcc)", 1, d_out);
    run<28>("v_cndmask_b32 (vcc), independent destination", 1, d_out);
    run<24>("v_cndmask_b32_e64 (sgpr pair mask)", 1, d_out);
    run<25>("v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (pair)", 2, d_out);
    run<26>("v_cmp_gt_f32 s[..] + v_cndmask_b32_e64 s[..] (pair)", 2, d_out);
    run<27>("v_addc_co_u32 (vcc in, vcc out)", 1, d_out);
    run<29>("v_cmp + s_and_saveexec + v_mov + s_mov exec (4 instr)", 4, d_out);
    run<30>("v_max_f32 + v_min_f32 (pair)", 2, d_out);
    run<31>("v_med3_f32", 1, d_out);
    run<32>("v_bfi_b32", 1, d_out);
    run<13>("v_cmp_gt_f32 -> vcc", 1, d_out);
    run<23>("v_add_co_u32 -> vcc", 1, d_out);
    run<14>("v_mov_b32_dpp row_shr:1", 1, d_out);
    run<21>("v_readlane_b32", 1, d_out);
    run<15>("v_log_f32", 1, d_out);
    run<8>("v_add_f64", 1, d_out);
    run<18>("v_mul_f64", 1, d_out);
    run<9>("v_fma_f64", 1, d_out);
    run<10>("v_cvt_f64_f32", 1, d_out);
    run<22>("v_cvt_f32_f64", 1, d_out);
    return 0;
}
