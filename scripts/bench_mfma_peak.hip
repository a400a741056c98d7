// Developer: the f32 MFMA rate the whole chip sustains (all 1024 SIMDs, 2 waves each, nothing but v_mfma_f32_32x32x2_f32),
// with the shader clock it runs at (s_memtime ticks per 100 MHz s_memrealtime tick).  The roofline peak of the guide is
// 157.3 TFLOP/s = 256 CUs x 256 flop/cycle x 2.4 GHz.
// build: hipcc -O3 --offload-arch=gfx950 scripts/bench_mfma_peak.hip -o build_tmp/bench_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int DEP>
__global__ void __launch_bounds__(256, 2) k(float* out, unsigned long long* clk, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (DEP) {          // the kernels' pattern: 16 dependent MFMAs per accumulator, then the next accumulator
#pragma unroll
                for (int v = 0; v < 4; ++v) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                continue;
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    const int G = 512;
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, G * 256 * 4); (void)hipMalloc(&c, G * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int pass = 0; pass < 4; ++pass) {
        const int iters = pass < 2 ? 20000 : 100000;
        const int dep = pass & 1;
        if (dep) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, d, c, iters); else hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, d, c, iters);
        (void)hipEventRecord(e0, 0);
        if (dep) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, d, c, iters); else hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, d, c, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2 * G]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double flop = (double)G * 4 * iters * 32.0 * 4096.0;
        printf("%s iters %6d: %8.3f ms  %6.1f TFLOP/s;  workgroup 0: %.1f s_memtime ticks per MFMA, s_memtime at %.0f MHz\n", dep ? "dependent  " : "independent", iters, ms,
               flop / ms * 1e-9, (double)h[0] / (iters * 32.0) / 2.0, (double)h[0] / (double)h[1] * 100.0);
    }
    return 0;
}
