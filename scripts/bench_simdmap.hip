// Developer: where do the waves of a persistent 512 x 256-thread launch land?  Prints, per workgroup, the XCC / SE / CU and
// the SIMD of each of its 4 waves (HW_ID), to decide whether the two workgroups of a CU put their wave 0 on the same SIMD.
// build: hipcc -O3 --offload-arch=gfx950 scripts/bench_simdmap.hip -o build_tmp/bench_simdmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256, 2) k(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;      // keep every workgroup resident while the rest launch
    lds[threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
    const int G = 512;
    unsigned* d; hipMalloc(&d, G * 4 * 2 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 76 * 1024, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 8); hipMemcpy(h.data(), d, G * 32, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;      // key (xcc, se, sh, cu) -> workgroups
    int bij = 0;
    for (int b = 0; b < G; ++b) {
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
        unsigned key = (xcc << 16) | (hw & 0xff00);
        cu[key].push_back(b);
        unsigned m = 0; for (int w = 0; w < 4; ++w) m |= 1u << ((h[(b * 4 + w) * 2] >> 4) & 3);
        bij += (m == 0xf);
    }
    printf("workgroups whose 4 waves sit on 4 distinct SIMDs: %d / %d; distinct CUs: %zu\n", bij, G, cu.size());
    int same0 = 0, pairs = 0, shown = 0;
    for (auto& kv : cu) {
        if (kv.second.size() != 2) continue;
        ++pairs;
        int a = kv.second[0], b = kv.second[1];
        unsigned sa = (h[a * 8] >> 4) & 3, sb = (h[b * 8] >> 4) & 3;
        same0 += (sa == sb);
        if (shown++ < 12) {
            printf("cu %06x: wg %3d simds", kv.first, a); for (int w = 0; w < 4; ++w) printf(" %u", (h[(a * 4 + w) * 2] >> 4) & 3);
            printf(" | wg %3d simds", b); for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
            printf("\n");
        }
    }
    printf("CUs with two workgroups: %d; wave 0 of both on the same SIMD: %d\n", pairs, same0);
    return 0;
}
