// experiments/probes/hwid_probe.hip — which CU does each workgroup of a 768-workgroup persistent grid land on?
//   hipcc --offload-arch=gfx950 -O3 -o hwid_probe hwid_probe.hip && ./hwid_probe
// Every workgroup records HW_REG_XCC_ID, HW_REG_HW_ID and its arrival order on its XCD, then idles ~200 us so that the whole
// grid is resident together (3 workgroups per CU by LDS, like fused64k_kernel).  Prints the raw HW_ID bit fields per XCD and, for
// several candidate field layouts, how many distinct CUs are seen and how many workgroups each got.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 3) void probe(unsigned* out, unsigned* ctr) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned r = atomicAdd(ctr + (xcc & 15), 1u);
        out[blockIdx.x * 4 + 0] = xcc;
        out[blockIdx.x * 4 + 1] = hw;
        out[blockIdx.x * 4 + 2] = r;
        lds[0] = (float)hw;
    }
    for (int i = 0; i < 3000; ++i) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
}
int main() {
    const int grid = 768;
    unsigned *d_out, *d_ctr;
    hipMalloc(&d_out, grid * 16);
    hipMalloc(&d_ctr, 64);
    hipMemset(d_ctr, 0, 64);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 34816, 0, d_out, d_ctr);
    std::vector<unsigned> h(grid * 4);
    hipMemcpy(h.data(), d_out, grid * 16, hipMemcpyDeviceToHost);
    unsigned all_or = 0, all_and = ~0u;
    for (int b = 0; b < grid; ++b) { all_or |= h[b * 4 + 1]; all_and &= h[b * 4 + 1]; }
    printf("HW_ID bits that vary: 0x%08x (or 0x%08x, and 0x%08x)\n", all_or & ~all_and, all_or, all_and);
    for (int x = 0; x < 8; ++x) {
        std::map<unsigned, std::vector<unsigned>> by_cu;   // key: HW_ID with wave / simd bits masked out
        int n = 0;
        for (int b = 0; b < grid; ++b)
            if ((h[b * 4] & 15) == (unsigned)x) { by_cu[h[b * 4 + 1] & ~0xFFu].push_back(h[b * 4 + 2]); ++n; }
        printf("XCC %d: %d workgroups on %zu distinct (HW_ID >> 8) values:", x, n, by_cu.size());
        for (auto& kv : by_cu) {
            printf(" [%03x:", kv.first >> 8);
            for (unsigned r : kv.second) printf(" %u", r);
            printf("]");
        }
        printf("\n");
    }
    printf("first 40 blocks (blockIdx: xcc hw_id>>8 arrival):");
    for (int b = 0; b < 40; ++b) printf(" %d:%u/%03x/%u", b, h[b * 4] & 15, h[b * 4 + 1] >> 8, h[b * 4 + 2]);
    printf("\n");
    return 0;
}
