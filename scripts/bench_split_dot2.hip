// Developer micro-benchmark: the three-plane bf16 split of fp32 registers with v_and_b32 + v_sub_f32 (11 VALU instructions per
// register pair, the shipped form) against v_dot2_f32_bf16 (residual = x - plane as a dot product of the packed plane with
// (-1, 0) / (0, -1) and x as the addend: 7 per pair).  Checks that both give the same planes bit for bit over random and edge-case
// inputs, and times both (one wave per SIMD and two).   hipcc --offload-arch=gfx950 -O3 scripts/bench_split_dot2.hip -o /tmp/bsd && /tmp/bsd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_and(unsigned x0, unsigned x1, unsigned (&P)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        P[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        if (p < 2) {
            x0 = __float_as_uint(__uint_as_float(x0) - __uint_as_float(x0 & 0xffff0000u));
            x1 = __float_as_uint(__uint_as_float(x1) - __uint_as_float(x1 & 0xffff0000u));
        }
    }
}
__device__ __forceinline__ void split_dot(unsigned x0, unsigned x1, unsigned (&P)[3]) {
    const bf16x2 m0 = __builtin_bit_cast(bf16x2, 0x0000bf80u), m1 = __builtin_bit_cast(bf16x2, 0xbf800000u);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        P[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        if (p < 2) {
            const bf16x2 a = __builtin_bit_cast(bf16x2, P[p]);
            x0 = __float_as_uint(__builtin_amdgcn_fdot2_f32_bf16(a, m0, __uint_as_float(x0), false));
            x1 = __float_as_uint(__builtin_amdgcn_fdot2_f32_bf16(a, m1, __uint_as_float(x1), false));
        }
    }
}

template <int MODE>
__global__ void check(const unsigned* in, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned P[3];
    if (MODE == 0) split_and(in[2 * i], in[2 * i + 1], P); else split_dot(in[2 * i], in[2 * i + 1], P);
    out[3 * i] = P[0]; out[3 * i + 1] = P[1]; out[3 * i + 2] = P[2];
}

template <int MODE>
__global__ void __launch_bounds__(512) rate(const unsigned* in, unsigned* out, int iters) {
    unsigned x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = in[threadIdx.x * 16 + j];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned P[3];
            if (MODE == 0) split_and(x[2 * j] + it, x[2 * j + 1] ^ it, P); else split_dot(x[2 * j] + it, x[2 * j + 1] ^ it, P);
            acc ^= P[0] + P[1] + P[2];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int n = 1 << 22;
    std::vector<unsigned> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float f;
        const int k = i & 15;
        if (k == 0) f = 0.f; else if (k == 1) f = -0.f; else if (k == 2) f = 1e-30f * (rand() / (float)RAND_MAX);
        else if (k == 3) f = 1e30f * (rand() / (float)RAND_MAX - 0.5f); else if (k == 4) f = 1.0f; else if (k == 5) f = -1.0f - 1.1920929e-7f;
        else f = (rand() / (float)RAND_MAX - 0.5f) * expf((rand() / (float)RAND_MAX - 0.5f) * 40.f);
        memcpy(&h[i], &f, 4);
    }
    unsigned *din, *d0, *d1;
    hipMalloc(&din, n * 4); hipMalloc(&d0, (size_t)n / 2 * 3 * 4); hipMalloc(&d1, (size_t)n / 2 * 3 * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    check<0><<<n / 2 / 256, 256>>>(din, d0, n);
    check<1><<<n / 2 / 256, 256>>>(din, d1, n);
    std::vector<unsigned> a((size_t)n / 2 * 3), b((size_t)n / 2 * 3);
    hipMemcpy(a.data(), d0, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
    printf("planes: %zu of %zu words differ", bad, a.size());
    if (bad) { float f0, f1; memcpy(&f0, &h[2 * (first / 3)], 4); memcpy(&f1, &h[2 * (first / 3) + 1], 4);
               printf(" (first: pair %zu plane %zu: and/sub %08x dot2 %08x, inputs %g %g)", first / 3, first % 3, a[first], b[first], f0, f1); }
    printf("\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int mode = 0; mode < 2; ++mode) {
            const int iters = 20000;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) rate<0><<<256, waves * 64>>>(din, d0, iters); else rate<1><<<256, waves * 64>>>(din, d0, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            // 8 pairs per iteration per wave
            printf("%d wave(s) per SIMD, %s: %.3f ms for %d x 8 pair splits per wave -> %.1f ns per pair split per wave\n", waves / 4,
                   mode ? "v_dot2_f32_bf16 (7 per pair)" : "v_and + v_sub   (11 per pair)", best, iters, best * 1e6 / (iters * 8.0));
        }
    }
    return 0;
}
