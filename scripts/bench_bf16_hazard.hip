// Developer: are v_mfma_f32_32x32x16_bf16 accumulation chains fed by planes split in registers deterministic at two waves
// per SIMD?  Every wave runs the same k-loop on the same data (the K^-1 k-loop of gpsat_kernels.hip: half blocks loaded,
// split into three bf16 planes, 24 MFMAs on four accumulators) with its own timing jitter; the host compares every wave's
// accumulators with wave 0's, bit for bit.
//   variant bit 0: products interleaved over the accumulators (else one chain of 6 per accumulator after the other)
//   variant bit 1: 64 idle cycles (s_nop) behind the MFMAs of a step
//   variant bit 2: sched_barrier between split and MFMAs and behind the MFMAs
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_bf16_hazard.hip -o /tmp/bench_hz && /tmp/bench_hz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct RawHalf { u32x4 q[2]; };
struct HalfPl { u32x4 p[3]; };

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
__device__ __forceinline__ void mma_half(f32x16& acc, const HalfPl& A, const HalfPl& B) {
    acc = mf(A.p[1], B.p[1], acc); acc = mf(A.p[0], B.p[2], acc); acc = mf(A.p[2], B.p[0], acc);
    acc = mf(A.p[0], B.p[1], acc); acc = mf(A.p[1], B.p[0], acc); acc = mf(A.p[0], B.p[0], acc);
}
struct Ops { RawHalf A0, A1, B0, B1; };
__device__ __forceinline__ void load(Ops& S, const float* ws, int step, int m, int lane) {
    S.A0 = ldh(ws, 4 * step, m, lane); S.A1 = ldh(ws, 4 * step + 1, m, lane);
    S.B0 = ldh(ws, 4 * step + 2, m, lane); S.B1 = ldh(ws, 4 * step + 3, m, lane);
}
template <int VAR>
__device__ __forceinline__ void comp(f32x16 (&acc)[4], const Ops& S) {
    const HalfPl A0 = split_half(S.A0), A1 = split_half(S.A1), B0 = split_half(S.B0), B1 = split_half(S.B1);
    if (VAR & 4) __builtin_amdgcn_sched_barrier(0);
    if (VAR & 1) {
#define PP(i, j) acc[0] = mf(A0.p[i], B0.p[j], acc[0]); acc[1] = mf(A0.p[i], B1.p[j], acc[1]); acc[2] = mf(A1.p[i], B0.p[j], acc[2]); acc[3] = mf(A1.p[i], B1.p[j], acc[3]);
        PP(1, 1) PP(0, 2) PP(2, 0) PP(0, 1) PP(1, 0) PP(0, 0)
#undef PP
    } else {
        mma_half(acc[0], A0, B0); mma_half(acc[1], A0, B1); mma_half(acc[2], A1, B0); mma_half(acc[3], A1, B1);
    }
    if (VAR & 4) __builtin_amdgcn_sched_barrier(0);
    if (VAR & 2) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
}
template <int VAR>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ ws, float* __restrict__ out, int nsteps, int reps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gw = blockIdx.x * 4 + w;
    f32x16 acc[4];
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        Ops S0, S1;
        load(S0, ws, 0, 0, lane);
        for (int s = 0; s < nsteps; ++s) {
            if (((s * 7 + gw * 13 + rep) & 15) == 0) __builtin_amdgcn_s_sleep(2);       // timing jitter, different per wave
            load(S1, ws, s, 1, lane);
            comp<VAR>(acc, S0);
            load(S0, ws, min(s + 1, nsteps - 1), 0, lane);
            comp<VAR>(acc, S1);
        }
        float* o = out + ((size_t)gw * reps + rep) * 4096;
        for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) o[(n * 16 + i) * 64 + lane] = acc[n][i];
    }
}

// ---- mixed: on every CU one workgroup runs fp32 chains (16 x v_mfma_f32_32x32x2_f32 per product, what the sweep does) and the
// other the bf16 k-loop above, so that each SIMD holds one wave of either kind.  kind = parity of the arrival count on the CU.
__device__ __forceinline__ f32x16 ldb(const float* ws, int blk, int lane) {
    RawHalf a = ldh(ws, blk, 0, lane), b = ldh(ws, blk, 1, lane);
    f32x16 v;
    for (int i = 0; i < 4; ++i) { v[i] = __uint_as_float(a.q[0][i]); v[4 + i] = __uint_as_float(a.q[1][i]); v[8 + i] = __uint_as_float(b.q[0][i]); v[12 + i] = __uint_as_float(b.q[1][i]); }
    return v;
}
__device__ __forceinline__ void mma_blk(f32x16& acc, const f32x16& a, const f32x16& b) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}
template <int VAR>
__global__ void __launch_bounds__(256, 2) kmix(const float* __restrict__ ws, float* __restrict__ out, int* __restrict__ kinds,
                                               int* __restrict__ census, int nsteps, int reps, int force_kind) {
    __shared__ int kind_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gw = blockIdx.x * 4 + w;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
        const int cu = (int)((xcc << 8) | ((hw >> 8) & 0xff));
        kind_s = force_kind >= 0 ? force_kind : (atomicAdd(&census[cu], 1) & 1);
    }
    __syncthreads();
    const int kind = kind_s;
    if (lane == 0) kinds[gw] = kind;
    f32x16 acc[4];
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        if (kind == 0) {
            f32x16 A0 = ldb(ws, 0, lane), A1 = ldb(ws, 1, lane), B0 = ldb(ws, 2, lane), B1 = ldb(ws, 3, lane);
            for (int s = 0; s < nsteps; ++s) {
                if (((s * 7 + gw * 13 + rep) & 15) == 0) __builtin_amdgcn_s_sleep(2);
                const int sn = min(s + 1, nsteps - 1);
                f32x16 nA0 = ldb(ws, 4 * sn, lane), nA1 = ldb(ws, 4 * sn + 1, lane), nB0 = ldb(ws, 4 * sn + 2, lane), nB1 = ldb(ws, 4 * sn + 3, lane);
                mma_blk(acc[0], A0, B0); mma_blk(acc[1], A0, B1); mma_blk(acc[2], A1, B0); mma_blk(acc[3], A1, B1);
                A0 = nA0; A1 = nA1; B0 = nB0; B1 = nB1;
            }
        } else {
            Ops S0, S1;
            load(S0, ws, 0, 0, lane);
            for (int s = 0; s < nsteps; ++s) {
                if (((s * 7 + gw * 13 + rep) & 15) == 0) __builtin_amdgcn_s_sleep(2);
                load(S1, ws, s, 1, lane);
                comp<VAR>(acc, S0);
                load(S0, ws, min(s + 1, nsteps - 1), 0, lane);
                comp<VAR>(acc, S1);
            }
        }
        float* o = out + ((size_t)gw * reps + rep) * 4096;
        for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) o[(n * 16 + i) * 64 + lane] = acc[n][i];
    }
}
int main() {
    const int nsteps = 12, reps = 4, nblk = 512, nw = nblk * 4;
    std::vector<float> h(4 * nsteps * 1024);
    srand(3);
    for (auto& v : h) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.3f;
    float *d, *o;
    (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o, (size_t)nw * reps * 4096 * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ho((size_t)nw * reps * 4096);
    for (int var = 0; var < 8; ++var) {
        for (int rnd = 0; rnd < 3; ++rnd) {
            switch (var) {
#define L(v) case v: hipLaunchKernelGGL(k<v>, dim3(nblk), dim3(256), 0, 0, d, o, nsteps, reps); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
            }
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0; double worst = 0;
            for (size_t i = 0; i < (size_t)nw * reps; ++i)
                if (memcmp(&ho[i * 4096], &ho[0], 4096 * 4)) {
                    ++bad;
                    for (int j = 0; j < 4096; ++j) { double e = fabs((double)ho[i * 4096 + j] - ho[j]); if (e > worst) worst = e; }
                }
            printf("variant %d (%s%s%s) round %d: %d of %d wave results differ from wave 0's, max abs diff %.3e\n", var,
                   var & 1 ? "interleaved" : "chained", var & 2 ? " +nops" : "", var & 4 ? " +sched_barrier" : "", rnd, bad, nw * reps, worst);
        }
    }
    int *kinds, *census;
    (void)hipMalloc(&kinds, nw * 4); (void)hipMalloc(&census, 4096 * 4);
    std::vector<int> hk(nw);
    for (int mode = 0; mode < 4; ++mode) {          // 0: mixed, chained bf16; 1: mixed, interleaved bf16 + nops; 2: all fp32; 3: all bf16
        for (int rnd = 0; rnd < 3; ++rnd) {
            (void)hipMemset(census, 0, 4096 * 4);
            const int force = mode == 2 ? 0 : mode == 3 ? 1 : -1;
            if (mode == 1) hipLaunchKernelGGL(kmix<3>, dim3(nblk), dim3(256), 0, 0, d, o, kinds, census, nsteps, reps, force);
            else hipLaunchKernelGGL(kmix<0>, dim3(nblk), dim3(256), 0, 0, d, o, kinds, census, nsteps, reps, force);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hk.data(), kinds, nw * 4, hipMemcpyDeviceToHost);
            for (int kind = 0; kind < 2; ++kind) {
                long ref = -1; int bad = 0, tot = 0; double worst = 0;
                for (int gwv = 0; gwv < nw; ++gwv) {
                    if (hk[gwv] != kind) continue;
                    for (int rep = 0; rep < reps; ++rep) {
                        const size_t i = (size_t)gwv * reps + rep;
                        if (ref < 0) ref = (long)i;
                        ++tot;
                        if (memcmp(&ho[i * 4096], &ho[(size_t)ref * 4096], 4096 * 4)) {
                            ++bad;
                            for (int j = 0; j < 4096; ++j) { double e = fabs((double)ho[i * 4096 + j] - ho[(size_t)ref * 4096 + j]); if (e > worst) worst = e; }
                        }
                    }
                }
                if (tot) printf("mixed mode %d round %d, %s waves: %d of %d results differ from the first, max abs diff %.3e\n", mode, rnd,
                                kind ? "bf16" : "fp32", bad, tot, worst);
            }
        }
    }
    return 0;
}
