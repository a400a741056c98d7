// Developer: what does one MFMA cost inside the kernels' k-loop (64 MFMAs on 4 accumulators per step, 4 operand blocks
// of 4 KiB streamed from the workgroup's own 1-MiB factor, register double buffering), 2 waves per SIMD on all CUs?
// modes: 0 = as in the kernels (64 v_mov per step), 1 = two steps per trip, operand sets swap roles (no copies),
//        2 = no loads at all (the MFMA ceiling), 3 = as 0 but every load hits the same 4 blocks (L1/L2 resident)
// build: hipcc -O3 --offload-arch=gfx950 scripts/bench_kloop.hip -o build_tmp/bench_kloop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BLK = 1024;
__device__ __forceinline__ f32x16 ldg(const float* __restrict__ ws, int blk, int lane) {
    const f32x4* q = reinterpret_cast<const f32x4*>(ws + (size_t)blk * BLK) + lane;
    f32x4 a = q[0], b = q[64], c = q[128], d = q[192];
    f32x16 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    r[8] = c[0]; r[9] = c[1]; r[10] = c[2]; r[11] = c[3]; r[12] = d[0]; r[13] = d[1]; r[14] = d[2]; r[15] = d[3];
    return r;
}
__device__ __forceinline__ void mma_blk(f32x16& acc, const f32x16& a, const f32x16& b) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ ws, float* out, unsigned long long* clk, int steps, int reps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* my = ws + (size_t)blockIdx.x * 256 * BLK;
    f32x16 W[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) W[n][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        // blocks of this wave: a strided walk through the workgroup's 256 blocks, different per wave and repetition
        const int base = (w * 61 + rep * 17) & 255;
        auto bl = [&](int kk, int j) { return MODE == 3 ? j : ((base + kk * 4 + j) & 255); };
        f32x16 A0 = ldg(my, bl(0, 0), lane), A1 = ldg(my, bl(0, 1), lane), B0 = ldg(my, bl(0, 2), lane), B1 = ldg(my, bl(0, 3), lane);
        if (MODE == 0 || MODE == 3) {
            for (int kk = 0; kk < steps; ++kk) {
                f32x16 nA0 = A0, nA1 = A1, nB0 = B0, nB1 = B1;
                if (kk + 1 < steps) { nA0 = ldg(my, bl(kk + 1, 0), lane); nA1 = ldg(my, bl(kk + 1, 1), lane); nB0 = ldg(my, bl(kk + 1, 2), lane); nB1 = ldg(my, bl(kk + 1, 3), lane); }
                mma_blk(W[0], A0, B0); mma_blk(W[1], A0, B1); mma_blk(W[2], A1, B0); mma_blk(W[3], A1, B1);
                A0 = nA0; A1 = nA1; B0 = nB0; B1 = nB1;
            }
        } else if (MODE == 1) {
            for (int kk = 0; kk + 1 < steps; kk += 2) {
                const f32x16 Y0 = ldg(my, bl(kk + 1, 0), lane), Y1 = ldg(my, bl(kk + 1, 1), lane), Z0 = ldg(my, bl(kk + 1, 2), lane), Z1 = ldg(my, bl(kk + 1, 3), lane);
                mma_blk(W[0], A0, B0); mma_blk(W[1], A0, B1); mma_blk(W[2], A1, B0); mma_blk(W[3], A1, B1);
                __builtin_amdgcn_sched_barrier(0);
                const int km = min(kk + 2, steps - 1);
                A0 = ldg(my, bl(km, 0), lane); A1 = ldg(my, bl(km, 1), lane); B0 = ldg(my, bl(km, 2), lane); B1 = ldg(my, bl(km, 3), lane);
                mma_blk(W[0], Y0, Z0); mma_blk(W[1], Y0, Z1); mma_blk(W[2], Y1, Z0); mma_blk(W[3], Y1, Z1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int kk = 0; kk < steps; ++kk) { mma_blk(W[0], A0, B0); mma_blk(W[1], A0, B1); mma_blk(W[2], A1, B0); mma_blk(W[3], A1, B1); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += W[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const float* ws, float* out, unsigned long long* clk, int steps, int reps, const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, ws, out, clk, steps, reps);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, ws, out, clk, steps, reps);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 512; ++i) avg += (double)h[i]; avg /= 512;
    const double mf = (double)steps * reps * 64;
    // two waves share a SIMD: wall cycles per MFMA of ONE wave / 2 = pipe cycles per MFMA
    printf("%-44s steps %2d: %7.2f ms  %6.1f TFLOP/s  %5.1f SIMD cycles per MFMA\n", name, steps, ms, 512.0 * 4 * mf * 4096.0 / ms * 1e-9, avg / mf / 2.0);
}
int main() {
    float* ws; float* out; unsigned long long* clk;
    (void)hipMalloc(&ws, (size_t)512 * 256 * BLK * 4); (void)hipMemset(ws, 0, (size_t)512 * 256 * BLK * 4);
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&clk, 512 * 8);
    for (int steps : {8, 14}) {
        const int reps = 4000 / steps;
        run<2>(ws, out, clk, steps, reps, "no loads (ceiling)");
        run<0>(ws, out, clk, steps, reps, "double buffer + 64 v_mov per step (kernels)");
        run<1>(ws, out, clk, steps, reps, "two steps per trip, no copies");
        run<3>(ws, out, clk, steps, reps, "as the kernels, loads hit 4 resident blocks");
    }
    return 0;
}
