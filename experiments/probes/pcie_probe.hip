// pcie_probe.hip — developer probe: pageable vs registered vs pinned host copies (1 GiB in, 0.5 GiB out).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t in_b = (size_t)1 << 30, out_b = (size_t)1 << 29;
    void *d_in, *d_out; CK(hipMalloc(&d_in, in_b)); CK(hipMalloc(&d_out, out_b));
    char* h_in = (char*)malloc(in_b); char* h_out = (char*)malloc(out_b);
    memset(h_in, 1, in_b); memset(h_out, 0, out_b);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpy(d_in, h_in, in_b, hipMemcpyHostToDevice)); CK(hipMemcpy(h_out, d_out, out_b, hipMemcpyDeviceToHost));
        double t1 = now();
        printf("pageable hipMemcpy H2D 1 GiB + D2H 0.5 GiB: %.1f ms\n", (t1 - t0) * 1e3);
        t0 = now();
        CK(hipHostRegister(h_in, in_b, hipHostRegisterDefault)); CK(hipHostRegister(h_out, out_b, hipHostRegisterDefault));
        t1 = now();
        CK(hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(h_in)); CK(hipHostUnregister(h_out));
        double t3 = now();
        printf("register %.1f ms, copies %.1f ms, unregister %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    void *p_in, *p_out; CK(hipHostMalloc(&p_in, in_b, 0)); CK(hipHostMalloc(&p_out, out_b, 0));
    double t0 = now(); memcpy(p_in, h_in, in_b); double t1 = now();
    CK(hipMemcpyAsync(d_in, p_in, in_b, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(p_out, d_out, out_b, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    double t2 = now(); memcpy(h_out, p_out, out_b); double t3 = now();
    printf("pinned: cpu memcpy in %.1f ms, dma %.1f ms, cpu memcpy out %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    return 0;
}
