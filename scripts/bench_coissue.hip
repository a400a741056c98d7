// Developer micro-benchmark: how two waves of one SIMD share it.  A 512-thread workgroup puts two waves on each SIMD
// (waves w and w+4); waves 0-3 run role A, waves 4-7 role B.  Roles: 0 idle, 1 back-to-back v_mfma_f32_32x32x2_f32 on 4
// accumulators, 2 independent v_fma_f32 chains (16), 3 a dependent v_fma_f32 chain, 4 exp-heavy VALU (K-block like).
// Prints cycles per role alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed) {
    if (ROLE == 1) {
        f32x16 a0, a1, a2, a3;
        for (int i = 0; i < 16; ++i) { a0[i] = seed; a1[i] = seed + 1; a2[i] = seed + 2; a3[i] = seed + 3; }
        float x = seed, y = seed * 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
        }
        return a0[0] + a1[1] + a2[2] + a3[3];
    } else if (ROLE == 2) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += v[i];
        return s;
    } else if (ROLE == 3) {
        float v = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 64; ++r) v = fmaf(v, 1.0001f, 0.5f);
        }
        return v;
    } else if (ROLE == 4) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + 0.01f * i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) { float d = v[i] - 0.3f; float r2 = fmaf(d, d, 0.1f); v[i] = __expf(-0.5f * r2) * 0.9f + 0.01f * v[i]; }
        }
        float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
        return s;
    }
    else if (ROLE == 5) {            // register moves (v_mov_b32 through inline asm so that they are not folded away)
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 15]));
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += v[i];
        return s;
    } else if (ROLE == 6) {          // transcendental only
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += v[i];
        return s;
    } else if (ROLE == 7) {          // integer / select ops
        int v[16];
        for (int i = 0; i < 16; ++i) v[i] = (int)seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 3) & 15]));
        }
        int s = 0; for (int i = 0; i < 16; ++i) s += v[i];
        return (float)s;
    } else if (ROLE == 8) {          // LDS reads
        extern __shared__ float lds[];
        float s = 0;
        const int l = threadIdx.x & 63;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s += lds[(l * 4 + r * 256 + it) & 4095];
        }
        return s;
    }
    else if (ROLE >= 10 && ROLE < 40) {       // one wave: MFMA with (ROLE - 10) independent v_fma fillers after each MFMA
        constexpr int NF = ROLE - 10;
        f32x16 a0, a1, a2, a3;
        for (int i = 0; i < 16; ++i) { a0[i] = seed; a1[i] = seed + 1; a2[i] = seed + 2; a3[i] = seed + 3; }
        float x = seed, y = seed * 0.5f;
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 15]) : "v"(y));
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 15]) : "v"(y));
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 15]) : "v"(y));
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 15]) : "v"(y));
            }
        }
        float sum = a0[0] + a1[1] + a2[2] + a3[3];
        for (int i = 0; i < 16; ++i) sum += v[i];
        return sum;
    }
    return seed;
}

template <int RA, int RB, int PA = 0, int PB = 0>
__global__ void __launch_bounds__(512) k(int iters, float* out, unsigned long long* cyc) {
    const int w = threadIdx.x >> 6;
    if (w < 4) __builtin_amdgcn_s_setprio(PA); else __builtin_amdgcn_s_setprio(PB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = (w < 4) ? run_role<RA>(iters, threadIdx.x * 1e-3f) : run_role<RB>(iters, threadIdx.x * 1e-3f);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}

template <int RA, int RB, int PA = 0, int PB = 0>
void go(const char* name, int iters, float* dout, unsigned long long* dc) {
    hipLaunchKernelGGL((k<RA, RB, PA, PB>), dim3(1), dim3(512), 16384, 0, iters, dout, dc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<RA, RB, PA, PB>), dim3(1), dim3(512), 16384, 0, iters, dout, dc);
    (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost);
    printf("%-40s A (wave 0): %9llu cycles   B (wave 4): %9llu cycles\n", name, c[0], c[4]);
}

int main() {
    float* dout; unsigned long long* dc;
    (void)hipMalloc(&dout, 4096); (void)hipMalloc(&dc, 64);
    const int N = 2000;
    go<1, 0>("mfma alone", N, dout, dc);
    go<0, 2>("16 indep. fma chains alone", N, dout, dc);
    go<0, 3>("dependent fma chain alone", N, dout, dc);
    go<0, 4>("exp-heavy valu alone", N, dout, dc);
    go<1, 1>("mfma + mfma", N, dout, dc);
    go<1, 2>("mfma + 16 indep. fma chains", N, dout, dc);
    go<1, 3>("mfma + dependent fma chain", N, dout, dc);
    go<1, 4>("mfma + exp-heavy valu", N, dout, dc);
    go<1, 2, 0, 1>("mfma(prio0) + indep fma (prio1)", N, dout, dc);
    go<1, 2, 0, 3>("mfma(prio0) + indep fma (prio3)", N, dout, dc);
    go<1, 4, 0, 2>("mfma(prio0) + exp valu (prio2)", N, dout, dc);
    go<2, 1, 1, 0>("indep fma (prio1, older) + mfma(prio0)", N, dout, dc);
    go<2, 1, 0, 0>("indep fma (older) + mfma", N, dout, dc);
    go<4, 1, 0, 0>("exp valu (older) + mfma", N, dout, dc);
    go<0, 5>("v_mov alone", N, dout, dc);
    go<1, 5>("mfma + v_mov", N, dout, dc);
    go<0, 6>("v_exp alone", N, dout, dc);
    go<1, 6>("mfma + v_exp", N, dout, dc);
    go<0, 7>("v_add_u32 alone", N, dout, dc);
    go<1, 7>("mfma + v_add_u32", N, dout, dc);
    go<0, 8>("lds reads alone", N, dout, dc);
    go<1, 8>("mfma + lds reads", N, dout, dc);
    go<10, 0>("mfma, 0 fillers per MFMA", N, dout, dc);
    go<12, 0>("mfma, 2 fillers per MFMA", N, dout, dc);
    go<14, 0>("mfma, 4 fillers per MFMA", N, dout, dc);
    go<18, 0>("mfma, 8 fillers per MFMA", N, dout, dc);
    go<22, 0>("mfma, 12 fillers per MFMA", N, dout, dc);
    go<26, 0>("mfma, 16 fillers per MFMA", N, dout, dc);
    go<18, 18>("both: mfma, 8 fillers per MFMA", N, dout, dc);
    go<18, 2>("mfma 8 fillers + indep fma", N, dout, dc);
    go<2, 2>("indep fma + indep fma", N, dout, dc);
    go<4, 4>("exp valu + exp valu", N, dout, dc);
    return 0;
}
