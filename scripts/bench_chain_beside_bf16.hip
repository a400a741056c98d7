// Developer (EXPERIMENTS.md E48): the diagonal chain's k-loop of gpsat_kernels.hip (three fp32 MFMA chains D00 += A0^T A0,
// D01 += A0^T A1, D11 += A1^T A1 per step, the forward-solve partial sums tp0 / tp1 += A[q] * z[..] as VALU FMAs with z read
// from LDS, operands double-buffered from memory, raised wave priority) beside a workgroup that runs the bf16 K^-1 k-loop
// (planes split in registers, 24 x v_mfma_f32_32x32x16_bf16 per half step) on the same SIMDs.  Every chain wave computes the
// same function of the same data over and over; any result that differs from the wave's first is counted, separately for
// the MFMA accumulators and the VALU sums, by lane quarter.
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_chain_beside_bf16.hip -o /tmp/bench_cb && /tmp/bench_cb [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct RawHalf { u32x4 q[2]; };
struct HalfPl { u32x4 p[3]; };
extern __shared__ float lds_f[];

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 ldg(const float* ws, int blk, int lane) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, 0x7fffffff, 0x00020000);
    const int so = blk * 4096, vo = lane * 16;
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo + 1024 * q, so, 16));
        v[4 * q] = a[0]; v[4 * q + 1] = a[1]; v[4 * q + 2] = a[2]; v[4 * q + 3] = a[3];
    }
    return v;
}
__device__ __forceinline__ RawHalf ldh(const float* ws, int blk, int m, int lane) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, 0x7fffffff, 0x00020000);
    const int so = blk * 4096 + m * 2048, vo = lane * 16;
    RawHalf h;
    h.q[0] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 16);
    h.q[1] = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 1024, so, 16);
    return h;
}
__device__ __forceinline__ HalfPl split_half(const RawHalf& v) {
    HalfPl P;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned x0 = v.q[j >> 1][2 * (j & 1)], x1 = v.q[j >> 1][2 * (j & 1) + 1];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            P.p[p][j] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
            if (p < 2) {
                x0 = __float_as_uint(__uint_as_float(x0) - __uint_as_float(x0 & 0xffff0000u));
                x1 = __float_as_uint(__uint_as_float(x1) - __uint_as_float(x1 & 0xffff0000u));
            }
        }
    }
    return P;
}
__device__ __forceinline__ f32x16 mf(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
struct Ops { RawHalf A0, A1, B0, B1; };
__device__ __forceinline__ void loadops(Ops& S, const float* ws, int step, int m, int lane) {
    S.A0 = ldh(ws, 4 * step, m, lane); S.A1 = ldh(ws, 4 * step + 1, m, lane);
    S.B0 = ldh(ws, 4 * step + 2, m, lane); S.B1 = ldh(ws, 4 * step + 3, m, lane);
}
__device__ __forceinline__ void comp(f32x16 (&acc)[4], const Ops& S) {
    const HalfPl A0 = split_half(S.A0), A1 = split_half(S.A1), B0 = split_half(S.B0), B1 = split_half(S.B1);
#define PP(i, j) acc[0] = mf(A0.p[i], B0.p[j], acc[0]); acc[1] = mf(A0.p[i], B1.p[j], acc[1]); acc[2] = mf(A1.p[i], B0.p[j], acc[2]); acc[3] = mf(A1.p[i], B1.p[j], acc[3]);
    PP(1, 1) PP(0, 2) PP(2, 0) PP(0, 1) PP(1, 0) PP(0, 0)
#undef PP
}
__device__ __forceinline__ void mma_blk(f32x16& acc, const f32x16& a, const f32x16& b) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}

// out: [0..3] VALU sums wrong by lane quarter (tp0), [4..7] (tp1), [8] MFMA accumulators wrong (lanes), [9] results checked
template <int PRIO>
__global__ void __launch_bounds__(256, 2) kmix(const float* __restrict__ ws, unsigned long long* out, int* __restrict__ census,
                                               int nsteps, int reps, int force_kind, float* sink) {
    __shared__ int kind_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gw = blockIdx.x * 4 + w, h = lane >> 5;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
        const int cu = (int)((xcc << 8) | ((hw >> 8) & 0xff));
        kind_s = force_kind >= 0 ? force_kind : (atomicAdd(&census[cu], 1) & 1);
    }
    for (int i = threadIdx.x; i < 32 * nsteps; i += 256) lds_f[i] = 0.25f + 0.001f * (float)(i % 97);
    __syncthreads();
    const int kind = kind_s;
    if (kind == 0) {
        unsigned long long bad0 = 0, bad1 = 0, badm = 0, checked = 0;
        float r0 = 0.f, r1 = 0.f;
        f32x16 R00, R01, R11;
        for (int rep = 0; rep < reps; ++rep) {
            f32x16 D00, D01, D11;
            for (int i = 0; i < 16; ++i) { D00[i] = 0.f; D01[i] = 0.f; D11[i] = 0.f; }
            float tp0 = 0.f, tp1 = 0.f;
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            f32x16 A0 = ldg(ws, 0, lane), A1 = ldg(ws, 1, lane);
            for (int k = 0; k < nsteps; ++k) {
                if (((k * 7 + gw * 13 + rep) & 15) == 0) __builtin_amdgcn_s_sleep(2);      // timing jitter
                const int kn = min(k + 1, nsteps - 1);
                const f32x16 nA0 = ldg(ws, 2 * kn, lane), nA1 = ldg(ws, 2 * kn + 1, lane);
                mma_blk(D00, A0, A0);
                mma_blk(D01, A0, A1);
                mma_blk(D11, A1, A1);
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    const float zk = lds_f[32 * k + rho(qq, h)];
                    tp0 = fmaf(A0[qq], zk, tp0);
                    tp1 = fmaf(A1[qq], zk, tp1);
                }
                A0 = nA0; A1 = nA1;
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (rep == 0) { r0 = tp0; r1 = tp1; R00 = D00; R01 = D01; R11 = D11; }
            else {
                ++checked;
                if (__float_as_uint(tp0) != __float_as_uint(r0)) ++bad0;
                if (__float_as_uint(tp1) != __float_as_uint(r1)) ++bad1;
                int m = 0;
                for (int i = 0; i < 16; ++i) m |= (__float_as_uint(D00[i]) != __float_as_uint(R00[i])) | (__float_as_uint(D01[i]) != __float_as_uint(R01[i])) |
                                                  (__float_as_uint(D11[i]) != __float_as_uint(R11[i]));
                badm += m;
            }
        }
        atomicAdd(&out[lane >> 4], bad0);
        atomicAdd(&out[4 + (lane >> 4)], bad1);
        atomicAdd(&out[8], badm);
        atomicAdd(&out[9], checked);
    } else {
        f32x16 acc[4];
        for (int rep = 0; rep < (reps * 5) / 2; ++rep) {        // a repetition takes half as long as a chain wave's
#pragma unroll
            for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
            Ops S0, S1;
            loadops(S0, ws, 0, 0, lane);
            for (int s = 0; s < nsteps; ++s) {
                if (((s * 7 + gw * 13 + rep) & 15) == 0) __builtin_amdgcn_s_sleep(2);
                loadops(S1, ws, s, 1, lane);
                comp(acc, S0);
                loadops(S0, ws, min(s + 1, nsteps - 1), 0, lane);
                comp(acc, S1);
            }
        }
        sink[(size_t)gw * 64 + lane] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    const int nsteps = 12, nblk = 512;
    std::vector<float> h(4 * nsteps * 1024);
    srand(3);
    for (auto& v : h) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.3f;
    float *d, *sink;
    unsigned long long* out;
    int* census;
    (void)hipMalloc(&d, h.size() * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&sink, (size_t)nblk * 4 * 64 * 4);
    (void)hipMalloc(&out, 128);
    (void)hipMalloc(&census, 4096 * 4);
    const char* names[4] = {"chain waves beside bf16 K^-1 waves", "chain waves only (both workgroups of a CU)", "chain waves beside bf16 waves, no s_setprio", "chain waves only, no s_setprio"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rnd = 0; rnd < 3; ++rnd) {
            (void)hipMemset(census, 0, 4096 * 4);
            (void)hipMemset(out, 0, 128);
            const int force = (mode & 1) ? 0 : -1;
            const size_t smem = 72 * 1024;
            if (mode < 2) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kmix<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                hipLaunchKernelGGL(kmix<1>, dim3(nblk), dim3(256), smem, 0, d, out, census, nsteps, reps, force, sink);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kmix<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                hipLaunchKernelGGL(kmix<0>, dim3(nblk), dim3(256), smem, 0, d, out, census, nsteps, reps, force, sink);
            }
            hipError_t e = hipDeviceSynchronize();
            unsigned long long ho[16];
            (void)hipMemcpy(ho, out, 128, hipMemcpyDeviceToHost);
            printf("%-48s round %d: lane-results checked %llu; tp0 wrong by lane quarter [%llu %llu %llu %llu]; tp1 wrong [%llu %llu %llu %llu]; MFMA accumulators wrong (lanes) %llu%s\n",
                   names[mode], rnd, ho[9], ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6], ho[7], ho[8], e == hipSuccess ? "" : "  HIP ERROR");
            fflush(stdout);
        }
    }
    return 0;
}
