// Developer micro-benchmark ("what would change the picture" item 3): one 32x32x32 block product
// S_A^T S_B on blocks in the accumulator layout, as
//   (a) 16 x v_mfma_f32_32x32x2_f32                              (what the kernels do: 1024 cycles per product and SIMD)
//   (b) fp32 operands split EXACTLY into three bf16 planes each (8 + 8 + 8 mantissa bits, by truncation), the six products
//       a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2 on v_mfma_f32_32x32x16_bf16 (12 MFMAs of 32 cycles), fp32 accumulation
//       (b1) planes made on the fly from the fp32 registers every product, (b2) planes kept in registers (pre-split storage)
//   (c) two planes, four products (16 mantissa bits)
// Prints cycles per product (one and two waves per SIMD) and the error of each variant against an fp64 product.
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_split_bf16.hip -o /tmp/bench_split && /tmp/bench_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Planes { u32x4 p[3][2]; };        // [plane][half of the block's 16 registers]: 8 bf16 per lane and half

__device__ __forceinline__ unsigned fbits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {      // the upper halves (bf16 by truncation) of two floats
    return __builtin_amdgcn_perm(fbits(hi), fbits(lo), 0x07060302u);
}
template <int NP>
__device__ __forceinline__ void split(const f32x16& v, Planes& P) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0 = v[8 * m + 2 * j], x1 = v[8 * m + 2 * j + 1];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                P.p[p][m][j] = pack_hi(x0, x1);
                x0 -= __uint_as_float(fbits(x0) & 0xffff0000u);
                x1 -= __uint_as_float(fbits(x1) & 0xffff0000u);
            }
        }
    }
}
__device__ __forceinline__ f32x16 mf(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int NP>
__device__ __forceinline__ void prod_planes(f32x16& acc, const Planes& A, const Planes& B) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (NP == 3) {       // smallest terms first
            acc = mf(A.p[1][m], B.p[1][m], acc);
            acc = mf(A.p[0][m], B.p[2][m], acc);
            acc = mf(A.p[2][m], B.p[0][m], acc);
        } else {
            acc = mf(A.p[1][m], B.p[1][m], acc);
        }
        acc = mf(A.p[0][m], B.p[1][m], acc);
        acc = mf(A.p[1][m], B.p[0][m], acc);
        acc = mf(A.p[0][m], B.p[0][m], acc);
    }
}

template <int VAR>
__global__ void __launch_bounds__(512) k(const float* __restrict__ Ain, const float* __restrict__ Bin, float* __restrict__ out,
                                         long long* __restrict__ cyc, int iters, int nwaves) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= nwaves) return;
    f32x16 A, B, acc;
    for (int i = 0; i < 16; ++i) { A[i] = Ain[i * 64 + lane]; B[i] = Bin[i * 64 + lane]; acc[i] = 0.f; }
    Planes PA, PB;
    if (VAR == 2) { split<3>(A, PA); split<3>(B, PB); }
    if (VAR == 4) { split<2>(A, PA); split<2>(B, PB); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], B[s], acc, 0, 0, 0);
        } else if (VAR == 1) {
            split<3>(A, PA); split<3>(B, PB);
            prod_planes<3>(acc, PA, PB);
        } else if (VAR == 2) {
            prod_planes<3>(acc, PA, PB);
        } else if (VAR == 3) {
            split<2>(A, PA); split<2>(B, PB);
            prod_planes<2>(acc, PA, PB);
        } else {
            prod_planes<2>(acc, PA, PB);
        }
        asm volatile("" : "+v"(A), "+v"(B));          // the operands are "new" every iteration: nothing is hoisted
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    if (blockIdx.x == 0 && w == 0) for (int i = 0; i < 16; ++i) out[i * 64 + lane] = acc[i] / (float)iters;
}

int main() {
    int iters = 2000;
    std::vector<float> hA(1024), hB(1024), hO(1024);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    // fp64 reference: element (row r, col m) of a block lives in register (r / 8) * 4 + r % 4 of lane m + 32 * ((r / 4) % 2)
    auto at = [](const std::vector<float>& b, int r, int m) { return (double)b[((r / 8) * 4 + r % 4) * 64 + m + 32 * ((r / 4) % 2)]; };
    std::vector<double> ref(1024);
    double scale = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int kk = 0; kk < 32; ++kk) s += at(hA, kk, i) * at(hB, kk, j);
        ref[((i / 8) * 4 + i % 4) * 64 + j + 32 * ((i / 4) % 2)] = s; scale = fmax(scale, fabs(s));
    }
    float *dA, *dB, *dO; long long* dC;
    (void)hipMalloc(&dA, 4096); (void)hipMalloc(&dB, 4096); (void)hipMalloc(&dO, 4096); (void)hipMalloc(&dC, 256 * 8 * 8);
    (void)hipMemcpy(dA, hA.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), 4096, hipMemcpyHostToDevice);
    const char* names[5] = {"fp32 16 x 32x32x2", "bf16 x 3 planes, split on the fly", "bf16 x 3 planes, pre-split", "bf16 x 2 planes, split on the fly", "bf16 x 2 planes, pre-split"};
    for (int var = 0; var < 5; ++var) {
        for (int nw = 4; nw <= 8; nw += 4) {
            auto launch = [&](int v) {
                switch (v) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, dA, dB, dO, dC, iters, nw); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, dA, dB, dO, dC, iters, nw); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, dA, dB, dO, dC, iters, nw); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, dA, dB, dO, dC, iters, nw); break;
                    default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, dA, dB, dO, dC, iters, nw); break;
                }
            };
            const int timed = iters;
            iters = 1; launch(var);                    // one product: its error (the timed run accumulates `iters` of them in fp32)
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(hO.data(), dO, 4096, hipMemcpyDeviceToHost);
            iters = timed; launch(var); launch(var);
            (void)hipDeviceSynchronize();
            std::vector<long long> hc(256 * 8);
            (void)hipMemcpy(hc.data(), dC, hc.size() * 8, hipMemcpyDeviceToHost);
            double c = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) c += (double)hc[b * 8 + w];
            c /= 256.0 * nw * iters;
            double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs((double)hO[i] - ref[i]));
            printf("%-36s %d waves/SIMD: %7.1f cycles per product and wave (%6.1f per SIMD), max error / max |result| = %.2e\n",
                   names[var], nw / 4, c, c / (nw / 4), err / scale);
        }
    }
    return 0;
}
