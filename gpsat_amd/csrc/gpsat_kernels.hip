// gpsat_kernels.hip -- gfx950 (MI355X, CDNA4) persistent local-expert exact-GP kernel.
//
// One 256-thread workgroup (4 wave64, one per SIMD) owns one expert tile from the first
// objective evaluation to the last prediction: kernel-matrix build, blocked Cholesky,
// triangular inverse, K^-1 contraction for the hyper-parameter gradient, the L-BFGS / Adam
// iteration and the predictive solves all run inside ONE launch; tiles are pulled from a
// cost-sorted queue with one atomic per tile.  Nothing is traced, nothing returns to the host
// between evaluations (the reference crosses host<->device per L-BFGS evaluation,
// GPSat/models/gpflow_models.py:317-321).
//
// Data layout ("acc layout").  Every 32x32 block lives in HBM/L2 exactly as the
// v_mfma_f32_32x32x2_f32 accumulator holds it: lane l = 32*h + g owns column g and the 16 rows
// rho(r,h) = (r&3) + 8*(r>>2) + 4*h, r = 0..15.  Register r of lane l is stored at float offset
// (r>>2)*256 + 4*l + (r&3), so a block is moved by four fully coalesced 1-KiB dwordx4
// wave-instructions and needs NO LDS staging and NO transposition:
//   * loaded as the A operand a stored block S acts as S^T, loaded as the B operand it acts as S
//     (both operands enumerate the contraction index in the same permuted order rho(s,h), s = MFMA
//     step, which is legal because the order of a sum is free);
//   * an accumulator is directly the B operand of the next MFMA chain.
// All three O(N^3) phases are written so that every product has the form  S_A^T * S_B:
//   potrf :  K = U^T U (U upper),  W_ji = K_ji - sum_{k<j} U_kj^T U_ki ;  U_ji = L_jj^-1 W_ji
//   trtri :  M = L^-1 (lower),     W_ij = sum_{k=j}^{i-1} U_ki^T M_kj   ;  M_ij = -L_ii^-1 W_ij
//   lauum :  K^-1 = M^T M,         (K^-1)_ab = sum_{c>=a} M_ca^T M_cb   (contracted in registers
//            against dK/dtheta recomputed on the fly, never stored)
// The maths follows SURVEY.md Appendix A; the objective is the reference's
// NLL = 1/2 y^T K^-1 y + sum log L_ii + N/2 log 2pi (GPSat/models/pure_python_gpr.py:485-487) and the
// gradient 1/2 sum Q .* dK/dtheta with Q = K^-1 - alpha alpha^T (ibid. :488-498).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gpsat_kernels.h"
#include "gpsat_ring.h"

// Diagnostic build only (-DGPSAT_PROFILE, scripts/phase_profile.py): per-wave cycle counters per code
// segment, accumulated in LDS and flushed to KernelArgs::prof.  No stamp executes in the product build.
#ifdef GPSAT_PROFILE
#define PROF_BEGIN() unsigned long long prof_t_ = __builtin_amdgcn_s_memtime()
#define PROF_END(c_, slot_)                                                                  \
    do {                                                                                     \
        unsigned long long t1_ = __builtin_amdgcn_s_memtime();                               \
        if ((c_).lane == 0) (c_).prof[(c_).w * 16 + (slot_)] += t1_ - prof_t_;               \
        prof_t_ = t1_;                                                                       \
    } while (0)
// event trace of ONE evaluation of workgroup 0 (scripts/phase_timeline.py): (cycle << 16) | (arg << 8) | code per wave
#define TRACE(c_, code_, arg_)                                                                        \
    do {                                                                                              \
        if ((c_).trace && (c_).lane == 0 && shared_state()->tron) {                                   \
            const int i_ = shared_state()->tcnt[(c_).w]++;                                            \
            if (i_ < 1024) (c_).trace[(c_).w * 1024 + i_] = (__builtin_amdgcn_s_memtime() << 16) |    \
                                                           ((unsigned long long)((arg_) & 255) << 8) | (code_); \
        }                                                                                             \
    } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_END(c_, slot_) do {} while (0)
#define TRACE(c_, code_, arg_) do {} while (0)
#endif

// Two builds of this file are linked: the default (4 waves per workgroup, two workgroups per CU when the tile's LDS
// allows) and -DGPSAT_W8 (8 waves, for batches whose largest tile needs more than half of the LDS, so that a CU still
// runs two waves per SIMD).  The variant lives in its own inner namespace; the entry points carry a suffix.
#ifdef GPSAT_W8
#define GPSAT_NW 8
#define GPSAT_VNS w8
#define GPSAT_VFN(name) name##_w8
#define GPSAT_MIN_WG 1
#else
#define GPSAT_VNS w4
#define GPSAT_VFN(name) name
#define GPSAT_MIN_WG 2
#endif

namespace gpsat {
namespace GPSAT_VNS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));       // v_pk_*_f32: two fp32 operations per VALU issue slot

// The whole LDS of a workgroup.  Everything in LDS is addressed as lds_f[offset] so that the compiler
// always knows the address space (ds_* instructions, never flat_*).
extern __shared__ __attribute__((aligned(16))) float lds_f[];

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int BLK = 1024;      // floats per block

// ---------------------------------------------------------------------------------------------
// block movement (acc layout) and the MFMA chain
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 pack16(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) {
    f32x16 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    r[8] = c[0]; r[9] = c[1]; r[10] = c[2]; r[11] = c[3];
    r[12] = d[0]; r[13] = d[1]; r[14] = d[2]; r[15] = d[3];
    return r;
}

// global block `blk` of the workgroup's workspace
#ifndef GPSAT_LD_AUX
#define GPSAT_LD_AUX 16          // cache policy of workspace block loads  (16 = sc1, 0 = default, 2 = nt)
#endif
#ifndef GPSAT_ST_AUX
#define GPSAT_ST_AUX 16          // ... and stores
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Workspace blocks move as buffer_load / buffer_store_dwordx4 with the block index in the scalar offset.
__device__ __forceinline__ f32x16 ldg(const float* __restrict__ ws, int blk, int lane) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, 0x7fffffff, 0x00020000);
    const int so = blk * (BLK * 4), vo = lane * 16;
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, GPSAT_LD_AUX));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo + 1024, so, GPSAT_LD_AUX));
    const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo + 2048, so, GPSAT_LD_AUX));
    const f32x4 d = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo + 3072, so, GPSAT_LD_AUX));
    return pack16(a, b, c, d);
}

__device__ __forceinline__ void stg(float* __restrict__ ws, int blk, int lane, const f32x16& v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
    const int so = blk * (BLK * 4), vo = lane * 16;
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    f32x4 c = {v[8], v[9], v[10], v[11]}, d = {v[12], v[13], v[14], v[15]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), r, vo, so, GPSAT_ST_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), r, vo + 1024, so, GPSAT_ST_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c), r, vo + 2048, so, GPSAT_ST_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), r, vo + 3072, so, GPSAT_ST_AUX);
}

// LDS block at float offset `off` (16-byte aligned)
__device__ __forceinline__ f32x16 ldl(int off, int lane) {
    const f32x4* q = reinterpret_cast<const f32x4*>(lds_f + off) + lane;
    return pack16(q[0], q[64], q[128], q[192]);
}

__device__ __forceinline__ void stl(int off, int lane, const f32x16& v) {
    f32x4* q = reinterpret_cast<f32x4*>(lds_f + off) + lane;
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    f32x4 c = {v[8], v[9], v[10], v[11]}, d = {v[12], v[13], v[14], v[15]};
    q[0] = a; q[64] = b; q[128] = c; q[192] = d;
}

// acc += S_A^T * S_B   (16 x v_mfma_f32_32x32x2_f32)
__device__ __forceinline__ void mma_blk(f32x16& acc, const f32x16& a, const f32x16& b) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}

// ---- an fp32 block product on the bf16 pipe, exactly (scripts/bench_split_bf16.hip) -- the K^-1 phase of both builds.
// An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 significand bits, by truncation); S_A^T S_B as the six plane
// products a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2 on v_mfma_f32_32x32x16_bf16 (fp32 accumulation; what is dropped is below
// 2^-24 of a product) has the error of the fp32 MFMA product (1.6e-7 against 2.0e-7 of max |result|) at less than half its
// cycles.  Registers 8m .. 8m+7 of a block in the accumulator layout are the 8-per-lane operand of that MFMA (the same k-slots
// in A and B); the planes are made in registers from the half block as it was loaded, once for the two products that use it.
// (Round 3 kept this loop out of the 4-wave build: beside it, 15 of 614 k evaluations of the CO-RESIDENT workgroup came out
// different from launch to launch.  Root cause, EXPERIMENTS.md E48: not this loop and no hand-off -- the factor was bit-identical
// in every one of 2 500 dumped events; what differed was ONE term of the diagonal chain's forward-solve sum, lost in lanes 48-63
// of the low half of a v_pk_fma_f32 that the SLP vectoriser had formed from the two scalar sums.  The fp32 kernels are built
// with -fno-slp-vectorize since; see the Makefile.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct RawHalf { u32x4 q[2]; };                  // registers 8m .. 8m+7 of a block, as loaded
struct HalfPl { u32x4 p[3]; };                   // their three bf16 planes: 8 bf16 per lane and plane

__device__ __forceinline__ RawHalf ldg_half(const float* __restrict__ ws, int blk, int m, int lane) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, 0x7fffffff, 0x00020000);
    const int so = blk * (BLK * 4) + m * 2048, vo = lane * 16;
    RawHalf h;
    h.q[0] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, GPSAT_LD_AUX);
    h.q[1] = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 1024, so, GPSAT_LD_AUX);
    return h;
}

__device__ __forceinline__ HalfPl split_half(const RawHalf& v) {
    HalfPl P;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned x0 = v.q[j >> 1][2 * (j & 1)], x1 = v.q[j >> 1][2 * (j & 1) + 1];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            P.p[p][j] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);                // the upper halves: bf16 by truncation
            if (p < 2) {
                x0 = __float_as_uint(__uint_as_float(x0) - __uint_as_float(x0 & 0xffff0000u));
                x1 = __float_as_uint(__uint_as_float(x1) - __uint_as_float(x1 & 0xffff0000u));
            }
        }
    }
    return P;
}

__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// All planes of a step are made before its first MFMA issues (the asm ties them), and eight idle wait states follow.  This is
// NOT a hazard workaround: the hardware needs one wait state between a VALU write and v_mfma_f32_32x32x16_bf16 reading the
// register (two for v_mfma_f32_32x32x2_f32) and hipcc always leaves two (scripts/bench_valu_mfma_hazard.hip: 0 stale reads in
// 1.3e9 trials per case from one wait state on).  It keeps the 24 MFMAs of a step back to back and costs nothing measurable
// (round 3's A/B); round 3 read its effect on the E48 flips as a missing wait state -- it only changed the partner's timing.
#define GPSAT_PLANES_SETTLE(...) asm volatile("s_nop 7" : __VA_ARGS__)
#define GPSAT_PL(P) "+v"((P).p[0]), "+v"((P).p[1]), "+v"((P).p[2])

// acc += (half of S_A)^T * (half of S_B) from planes (6 MFMAs), the smallest terms first
__device__ __forceinline__ void mma_half(f32x16& acc, const HalfPl& A, const HalfPl& B) {
    acc = mfma_bf(A.p[1], B.p[1], acc);
    acc = mfma_bf(A.p[0], B.p[2], acc);
    acc = mfma_bf(A.p[2], B.p[0], acc);
    acc = mfma_bf(A.p[0], B.p[1], acc);
    acc = mfma_bf(A.p[1], B.p[0], acc);
    acc = mfma_bf(A.p[0], B.p[0], acc);
}

// The k-loop of a K^-1 group on the bf16 pipe: half blocks (16 k-rows) per step, the halves of step h + 1 in
// flight while the halves of step h are split into planes and multiplied (24 MFMAs; 18 on the diagonal, where B = A and the
// upper product is not needed).  Two operand sets in ping-pong (m = 0 / m = 1 of a block row): no register copies.
struct KinvOps { RawHalf A0, A1, B0, B1; };

template <bool DIAG>
__device__ __forceinline__ void kinv_comp(f32x16 (&acc)[4], const KinvOps& S) {
    HalfPl A0 = split_half(S.A0), A1 = split_half(S.A1);
    if (DIAG) {
        GPSAT_PLANES_SETTLE(GPSAT_PL(A0), GPSAT_PL(A1));
#define GPSAT_PP(i, j)                              \
        acc[0] = mfma_bf(A0.p[i], A0.p[j], acc[0]); \
        acc[2] = mfma_bf(A1.p[i], A0.p[j], acc[2]); \
        acc[3] = mfma_bf(A1.p[i], A1.p[j], acc[3]);
        GPSAT_PP(1, 1) GPSAT_PP(0, 2) GPSAT_PP(2, 0) GPSAT_PP(0, 1) GPSAT_PP(1, 0) GPSAT_PP(0, 0)
#undef GPSAT_PP
    } else {
        HalfPl B0 = split_half(S.B0), B1 = split_half(S.B1);
        GPSAT_PLANES_SETTLE(GPSAT_PL(A0), GPSAT_PL(A1), GPSAT_PL(B0), GPSAT_PL(B1));
#define GPSAT_PP(i, j)                              \
        acc[0] = mfma_bf(A0.p[i], B0.p[j], acc[0]); \
        acc[1] = mfma_bf(A0.p[i], B1.p[j], acc[1]); \
        acc[2] = mfma_bf(A1.p[i], B0.p[j], acc[2]); \
        acc[3] = mfma_bf(A1.p[i], B1.p[j], acc[3]);
        GPSAT_PP(1, 1) GPSAT_PP(0, 2) GPSAT_PP(2, 0) GPSAT_PP(0, 1) GPSAT_PP(1, 0) GPSAT_PP(0, 0)
#undef GPSAT_PP
    }
}


__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ float xhalf_sum(float v) {   // v(lane) + v(lane ^ 32)
    return v + __shfl_xor(v, 32);
}

__device__ __forceinline__ void wave_lds_sync() {       // LDS is in-order per wave: only the compiler must not reorder
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// one wave pulls the next work index from an LDS counter (wave-uniform result)
__device__ __forceinline__ int wave_pull(int* counter, int lane) {
    int v = 0;
    if (lane == 0) v = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------------------------
// covariance functions.  r2 is the squared scaled distance.
//   kf = k(r), gg = g(r) with dk/dl_d = g(r) (x_d-x'_d)^2 / l_d^3   (both without sigma_f^2)
// f32 MFMA and every other vector instruction take turns on the SIMD (scripts/bench_coissue.hip: nothing issues beside
// a v_mfma_f32_32x32x2_f32, ~5 cycles per VALU instruction on top of 64 per MFMA), so each instruction per matrix
// element saved here is paid back in full.  RBF: the coordinates in LDS carry the factor KSC = sqrt(log2(e) / 2) on top
// of 1 / l, so that exp(-r^2 / 2) is ONE v_exp_f32 of the accumulated squared distance.
// ---------------------------------------------------------------------------------------------
template <int KERN>
struct KScale {                            // factor folded into the scaled coordinates, and its square
    static constexpr float c = (KERN == 0) ? 0.8493218002880191f : 1.0f;
    static constexpr float c2 = (KERN == 0) ? 0.7213475204444817f : 1.0f;
};

template <int KERN>
__device__ __forceinline__ void kfun(float r2, float& kf, float& gg) {
    if (KERN == 0) {                       // RBF: exp(-r^2/2) = 2^(-r2) on the prescaled distance
        kf = __builtin_amdgcn_exp2f(-r2);
        gg = kf;
    } else {
        float r = sqrtf(fmaxf(r2, 1e-36f));
        if (KERN == 1) {                   // Matern-1/2: exp(-r);  g = exp(-r)/r
            kf = __expf(-r);
            gg = kf / r;
        } else if (KERN == 2) {            // Matern-3/2: (1+s)exp(-s), s = sqrt3 r;  g = 3 exp(-s)
            float s = 1.7320508075688772f * r;
            float e = __expf(-s);
            kf = (1.f + s) * e;
            gg = 3.f * e;
        } else {                           // Matern-5/2: (1+s+s^2/3)exp(-s), s = sqrt5 r; g = 5/3 (1+s) exp(-s)
            float s = 2.23606797749979f * r;
            float e = __expf(-s);
            kf = (1.f + s + s * s * (1.f / 3.f)) * e;
            gg = (5.f / 3.f) * (1.f + s) * e;
        }
    }
}

#define GPSAT_PT_MAXNB 100      // block columns of the largest tile (gpsat_max_tile_obs: 3168 = 99 * 32)
#include "gpsat_opt.h"
#include "gpsat_coop.h"

constexpr int SHARED_FLOATS = (int)((sizeof(Shared) + 15) / 16) * 4;

__device__ __forceinline__ Shared* shared_state() { return reinterpret_cast<Shared*>(lds_f); }

// float offsets into lds_f
struct Lay { int xs, xsc, y, z, alpha, Ad, LT, U01, Wh, tmp, piv; };

template <int D, int KN>
struct Ctx {
    Lay L;
    float* ws;                   // this workgroup's global workspace
    int zb, dT0, vs0, cv0;       // block indices: zero block, DinvT[0], V scratch, V of all chunks (full covariance)
    int gp0;                     // byte offset of the gradient phase's per-group partial sums (aliases the V scratch)
    // cooperative tiles (gpsat_coop.h): the control block of the tile's OWNER, z and alpha of the open evaluation in the
    // owner's workspace, and whether this workgroup is a helper of that owner
    gCoopCtl* ctl;
    gfloat* zg;
    gfloat* ag;
    bool helper;
    int N, NB, Npad, P;
    int tid, lane, w, h, g;
    float sf2, sn2;
    unsigned long long* prof;    // LDS, [NW][16] (diagnostic build)
    unsigned long long* trace;   // global, [NW][1024] (diagnostic build, workgroup 0 only)
};

// ---------------------------------------------------------------------------------------------
// K block (rows 32*bi.., cols 32*bj..) in acc layout, built from the scaled coordinates in LDS.
// Padding rows/cols (index >= N) form an identity block so that they add exactly 0 to the log-det,
// the quadratic form and the gradient ("padding identity").
// ---------------------------------------------------------------------------------------------
template <int D, int KERN>
__device__ __forceinline__ f32x16 kblock_t(const Ctx<D, KERN>& c, int bi, int bj) {
    const int q = 32 * bj + c.g;
    float xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = lds_f[c.L.xsc + d * c.Npad + q];
    f32x16 out;
    if (bi != bj && 32 * bi + 32 <= c.N && 32 * bj + 32 <= c.N) {
        // interior off-diagonal block (most of them): no padding, no diagonal -- nothing to mask
        // row pairs in packed fp32 (same operations and rounding as the scalar form, half the issue slots)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int p0 = 32 * bi + 8 * qq + 4 * c.h;
            f32x2 r2[2];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const f32x4 xp = *reinterpret_cast<const f32x4*>(lds_f + c.L.xsc + d * c.Npad + p0);
                const f32x2 xb = {xq[d], xq[d]};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x2 xa = {xp[2 * e], xp[2 * e + 1]};
                    const f32x2 df = xa - xb;
                    r2[e] = (d == 0) ? df * df : __builtin_elementwise_fma(df, df, r2[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float kf, gg;
                kfun<KERN>(r2[e >> 1][e & 1], kf, gg);
                out[4 * qq + e] = c.sf2 * kf;
            }
        }
        return out;
    }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * bi + 8 * qq + 4 * c.h;
        float r2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            f32x4 xp = *reinterpret_cast<const f32x4*>(lds_f + c.L.xsc + d * c.Npad + p0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { float df = xp[e] - xq[d]; r2[e] = fmaf(df, df, r2[e]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = p0 + e;
            float kf, gg;
            kfun<KERN>(r2[e], kf, gg);
            float v = c.sf2 * kf;
            const bool valid = (p < c.N) && (q < c.N);
            v = valid ? v : 0.f;
            if (p == q) v = (p < c.N) ? (v + c.sn2) : 1.f;
            out[4 * qq + e] = v;
        }
    }
    return out;
}

template <int D, int KN>
__device__ __forceinline__ f32x16 kblock(const Ctx<D, KN>& c, int bi, int bj) {
    return kblock_t<D, KN>(c, bi, bj);
}

// cross-covariance block: rows = observations 32*bj.., cols = prediction points (xq scaled)
template <int D, int KERN>
__device__ __forceinline__ f32x16 ksblock_t(const Ctx<D, KERN>& c, int bj, const float (&xq)[D], bool qvalid) {
    f32x16 out;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * bj + 8 * qq + 4 * c.h;
        float r2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            f32x4 xp = *reinterpret_cast<const f32x4*>(lds_f + c.L.xsc + d * c.Npad + p0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { float df = xp[e] - xq[d]; r2[e] = fmaf(df, df, r2[e]); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float kf, gg;
            kfun<KERN>(r2[e], kf, gg);
            out[4 * qq + e] = ((p0 + e < c.N) && qvalid) ? c.sf2 * kf : 0.f;
        }
    }
    return out;
}

template <int D, int KN>
__device__ __forceinline__ f32x16 ksblock(const Ctx<D, KN>& c, int bj, const float (&xq)[D], bool qvalid) {
    return ksblock_t<D, KN>(c, bj, xq, qvalid);
}

// ---------------------------------------------------------------------------------------------
// gradient contraction of one (K^-1)_ab block held in registers.
//   acc_l[d] += wgt * Q * g(r) * (scaled diff_d)^2 ; acc_sf += wgt * Q * kf ; acc_sn += Q on the diagonal
// ---------------------------------------------------------------------------------------------
template <int D, int KERN>
__device__ __forceinline__ void contract_t(const Ctx<D, KERN>& c, const f32x16& kinv, int ba, int bb, float wgt,
                                           float (&accl)[D], float& accsf, float& accsn) {
    const int q = 32 * bb + c.g;
    float xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = lds_f[c.L.xsc + d * c.Npad + q];
    const float aq = lds_f[c.L.alpha + q];
    const bool qv = q < c.N;
    // all LDS operands of the block first (independent reads in flight together), then the arithmetic
    f32x4 xp[4][D], ap[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * ba + 8 * qq + 4 * c.h;
#pragma unroll
        for (int d = 0; d < D; ++d) xp[qq][d] = *reinterpret_cast<const f32x4*>(lds_f + c.L.xsc + d * c.Npad + p0);
        ap[qq] = *reinterpret_cast<const f32x4*>(lds_f + c.L.alpha + p0);
    }
    if (ba != bb && 32 * ba + 32 <= c.N && 32 * bb + 32 <= c.N) {
        // interior off-diagonal block: no masks, no diagonal term; the block weight is applied once to the block's sums
        // row pairs in packed fp32; even and odd rows keep separate partial sums
        f32x2 bl[D], bsf = {0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) bl[d] = f32x2{0.f, 0.f};
        const f32x2 naq = {-aq, -aq};
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                f32x2 d2[D];
                f32x2 r2 = {0.f, 0.f};
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const f32x2 xa = {xp[qq][d][2 * e], xp[qq][d][2 * e + 1]};
                    const f32x2 xb = {xq[d], xq[d]};
                    const f32x2 df = xa - xb;
                    d2[d] = df * df;
                    r2 = (d == 0) ? d2[d] : r2 + d2[d];
                }
                float kf0, gg0, kf1, gg1;
                kfun<KERN>(r2[0], kf0, gg0);
                kfun<KERN>(r2[1], kf1, gg1);
                const f32x2 kf = {kf0, kf1}, gg = {gg0, gg1};
                const f32x2 apv = {ap[qq][2 * e], ap[qq][2 * e + 1]};
                const f32x2 kv = {kinv[4 * qq + 2 * e], kinv[4 * qq + 2 * e + 1]};
                const f32x2 Q = __builtin_elementwise_fma(apv, naq, kv);
                bsf = __builtin_elementwise_fma(Q, kf, bsf);
                const f32x2 wg = Q * gg;
#pragma unroll
                for (int d = 0; d < D; ++d) bl[d] = __builtin_elementwise_fma(wg, d2[d], bl[d]);
            }
        }
        accsf = fmaf(wgt, bsf[0] + bsf[1], accsf);
#pragma unroll
        for (int d = 0; d < D; ++d) accl[d] = fmaf(wgt, bl[d][0] + bl[d][1], accl[d]);
        return;
    }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * ba + 8 * qq + 4 * c.h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = p0 + e;
            float d2[D];
            float r2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float df = xp[qq][d][e] - xq[d];
                d2[d] = df * df;
                r2 += d2[d];
            }
            float kf, gg;
            kfun<KERN>(r2, kf, gg);
            float Q = kinv[4 * qq + e] - ap[qq][e] * aq;
            Q = (qv && p < c.N) ? Q : 0.f;
            const float wq = wgt * Q;
            accsf = fmaf(wq, kf, accsf);
            const float wg = wq * gg;
#pragma unroll
            for (int d = 0; d < D; ++d) accl[d] = fmaf(wg, d2[d], accl[d]);
            if (p == q) accsn += Q;
        }
    }
}

template <int D, int KN>
__device__ __forceinline__ void contract(const Ctx<D, KN>& c, const f32x16& kinv, int ba, int bb, float wgt,
                                         float (&accl)[D], float& accsf, float& accsn) {
    contract_t<D, KN>(c, kinv, ba, bb, wgt, accl, accsf, accsn);
}

// ---------------------------------------------------------------------------------------------
// 32x32 diagonal block: X with X W X^T = I ("L^-1"; only X and X^T are ever used, X need not be
// triangular), one wave.
// Block Gaussian elimination with 2x2 pivots (W is SPD, no pivoting) of the augmented [W | I]:
// W = Lb Db Lb^T, [W | I] -> [.. | Lb^-1], then X = Cb^-1 Lb^-1 with Db = Cb Cb^T (2x2 Cholesky per pivot
// block).  Both 32x32 tiles stay in the MFMA accumulator layout and every elimination step is ONE rank-2
// update per tile, issued as v_mfma_f32_32x32x2_f32:  T -= M[32x2] * R[2x32].
//   * R (the two pivot rows) sits in registers r0, r0+1 of the half that owns rows 2s, 2s+1 -- already
//     "column on lane", i.e. directly the B operand (the other half fetches it with one v_permlane32_swap);
//   * M (the multipliers of every row) needs the two pivot COLUMNS; the trailing matrix is symmetric, so
//     they equal the pivot rows and are again available per lane.
// No LDS round trip, no cross-lane reduction inside the 16 dependent steps.
// In : W (acc layout).  Out: S1 = X (acc layout), S2 = X^T (acc layout, one transposition through the LDS
//      scratch Ad), logsum = -log|det X| (fp64), bad = 1 when a pivot block is not positive definite (or NaN).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_xor32(float x, int h) {
    // value of lane ^ 32 (v_permlane32_swap: VALU, no LDS round trip)
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(h ? r[0] : r[1]);
}

__device__ __forceinline__ void diag_factor(const f32x16& W, int Ad, int piv, int lane, f32x16& S1, f32x16& S2,
                                            double& logsum, int& bad) {
    const int h = lane >> 5, g = lane & 31;
    f32x16 TA = W;
    f32x16 TE;
#pragma unroll
    for (int r = 0; r < 16; ++r) TE[r] = (rho(r, h) == g) ? 1.f : 0.f;
    int isbad = 0;
    float q00 = 1.f, q10 = 0.f, q11 = 1.f;            // lane s (< 16) keeps pivot block s
    // 16 dependent steps; the critical chain per step is MFMA -> 3 readlanes -> det, rcp -> multipliers -> MFMA.
    // Everything that is not on it (2x2 Cholesky, scaling of the finished rows, log) is done once afterwards.
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k0 = 2 * s, k1 = k0 + 1;
        const int hp = (k0 >> 2) & 1;                 // half owning rows k0, k1
        const int r0 = (k0 & 3) + 4 * (k0 >> 3);      // their registers: r0, r0 + 1
        const int r1 = r0 + 1;
        float p00 = readlane_f(TA[r0], 32 * hp + k0);
        float p10 = readlane_f(TA[r1], 32 * hp + k0);
        float p11 = readlane_f(TA[r1], 32 * hp + k1);
        float det = p00 * p11 - p10 * p10;
        if (!(p00 > 0.f) || !(det > 0.f)) { isbad = 1; p00 = 1.f; p10 = 0.f; p11 = 1.f; det = 1.f; }
        float rd = __builtin_amdgcn_rcpf(det);
        rd = rd * (2.f - det * rd);
        const bool mine = lane == s;
        q00 = mine ? p00 : q00;
        q10 = mine ? p10 : q10;
        q11 = mine ? p11 : q11;
        // pivot rows as seen by this lane's column g (own registers in half hp, swapped in otherwise)
        const bool own = (h == hp);
        const float a0 = TA[r0], a1 = TA[r1], e0 = TE[r0], e1 = TE[r1];
        const float xa0 = lane_xor32(a0, h), xa1 = lane_xor32(a1, h);
        const float xe0 = lane_xor32(e0, h), xe1 = lane_xor32(e1, h);
        const float w0 = own ? a0 : xa0, w1 = own ? a1 : xa1;       // W'[k0][g], W'[k1][g]  (= W'[g][k0], W'[g][k1])
        const float f0 = own ? e0 : xe0, f1 = own ? e1 : xe1;       // E[k0][g],  E[k1][g]
        // multipliers of row g:  [m0 m1] = [w0 w1] P^-1, rows <= k1 are finished
        float m0 = (w0 * p11 - w1 * p10) * rd;
        float m1 = (w1 * p00 - w0 * p10) * rd;
        const bool below = g > k1;
        m0 = below ? m0 : 0.f;
        m1 = below ? m1 : 0.f;
        const float am = h ? -m1 : -m0;               // A operand: lane (h, i) supplies A[i][h]
        const float bA = h ? w1 : w0;                 // B operand: lane (h, j) supplies R[h][j]
        const float bE = h ? f1 : f0;
        TA = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bA, TA, 0, 0, 0);
        TE = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bE, TE, 0, 0, 0);
    }
    // TE = Lb^-1.  2x2 Cholesky of every pivot block in parallel (lane s owns block s):
    // C = [[c00,0],[c10,c11]], C^-1 = [[1/c00,0],[-c10/(c00 c11),1/c11]]; coefficients -> LDS (piv) -> all lanes
    {
        const float c00 = sqrtf(q00);
        const float c10 = q10 / c00;
        const float c11 = sqrtf(q11 - c10 * c10);
        const float i00 = 1.0f / c00, i11 = 1.0f / c11;
        const float i10 = -c10 * i00 * i11;
        if (lane < 16) {
            lds_f[piv + lane] = i00;
            lds_f[piv + 16 + lane] = i10;
            lds_f[piv + 32 + lane] = i11;
        }
        // -log|det X| = sum log(c00 c11) over pivot blocks, fp64
        double lg = (lane < 16) ? log((double)c00 * (double)c11) : 0.0;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) lg += __shfl_xor(lg, off);
        logsum = __shfl(lg, 0);
    }
    wave_lds_sync();
    // X rows: even row k0 -> i00 * E[k0];  odd row k1 -> i10 * E[k0] + i11 * E[k1]; registers (2t, 2t+1) of a
    // lane are the rows (k0, k1) of pivot block rho(2t, h) / 2
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int blk = rho(2 * t, h) >> 1;
        const float i00 = lds_f[piv + blk], i10 = lds_f[piv + 16 + blk], i11 = lds_f[piv + 32 + blk];
        const float e0 = TE[2 * t], e1 = TE[2 * t + 1];
        S1[2 * t] = i00 * e0;
        S1[2 * t + 1] = fmaf(i10, e0, i11 * e1);
    }
    // X^T in acc layout: one transposition through LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_f[Ad + rho(r, h) * 33 + g] = S1[r];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) S2[r] = lds_f[Ad + g * 33 + rho(r, h)];
    bad = isbad;
}

// ---------------------------------------------------------------------------------------------
// phase PT: blocked Cholesky K = U^T U (upper) and, in the same sweep, M = L^-1 (lower), processed
// by 2-row panels (rows j0, j0+1).  A panel's work items are block columns: U-type i > j1 and (when
// the gradient is wanted) M-type c < j0; a wave processes groups of 2 columns x 2 rows (4
// accumulators) with register double-buffering: the 4 blocks of step k+1 are in flight while the 64
// MFMAs of step k issue.  The panel blocks U_k,j0 / U_k,j1 are common to all waves (L1/L2 hits).
//   U-type column i > j1 : W_r = K_jr,i - sum_{k<j0} U_k,jr^T U_ki
//   M-type column c < j0 : W_r =        - sum_{k=c}^{j0-1} U_k,jr^T M_kc        (M_cc = L_cc^-1)
//   row j0 : X_j0 = L_j0j0^-1 W_0 ;  row j1 : X_j1 = L_j1j1^-1 (W_1 - U_j0j1^T X_j0)
// Wave 0 owns the 2x2 diagonal part (both 32x32 factorisations), the forward solve z = L^-1 y and the
// block M_j1,j0.  alpha = M^T z is accumulated as the M blocks are produced.
// Storage: U_ki (k < i) in the upper half of an NB x NB square of blocks, M_ik (i > k) in the lower
// half, M_kk = L_kk^-1 on the diagonal; (L_kk^-1)^T in a side array (DinvT).
// ---------------------------------------------------------------------------------------------
template <int D>
struct Panel {
    int j0, j1, has1, nU, nItems;
};

template <int D>
__device__ __forceinline__ int pt_item_col(const Panel<D>& p, int e) { return (e < p.nU) ? (p.j1 + 1 + e) : (e - p.nU); }

// k-loop of the group holding items 2g, 2g+1: on return W[2r+n] is the finished right-hand side of
// row jr, item n.
template <int D, int KN>
__device__ __forceinline__ void pt_group_kloop(const Ctx<D, KN>& c, const Panel<D>& p, int g, f32x16 (&W)[4]) {
    const int NB = c.NB, lane = c.lane, j0 = p.j0;
    const int e0 = 2 * g, e1 = e0 + 1;
    const bool v0 = e0 < p.nItems, v1 = e1 < p.nItems;
    const bool u0 = e0 < p.nU, u1 = e1 < p.nU;
    const int c0 = v0 ? pt_item_col(p, e0) : 0, c1 = v1 ? pt_item_col(p, e1) : 0;
    const int ks0 = v0 ? (u0 ? 0 : c0) : j0, ks1 = v1 ? (u1 ? 0 : c1) : j0;
    const int kmin = min(ks0, ks1);
#pragma unroll
    for (int n = 0; n < 4; ++n) W[n] = zero16();
    // Exact three-plane bf16 products (see mma_half / kinv_comp): half blocks of 16 k-rows per step, the halves of the next
    // steps in flight, operand sets in rotation (no register copies), 24 v_mfma_f32_32x32x16_bf16 per half step where
    // the fp32 pipe took 32 v_mfma_f32_32x32x2_f32 of twice the length (round 4: configs[1] 55.5 -> 61 % of the fp32 peak,
    // parity bounds unchanged).  The rows of a block column arrive in order, so a loop may start before its panel's chain is done.
    if (kmin < j0) {
        const int a1b = p.has1 ? 1 : 0;
        auto load = [&](KinvOps& S, int k, int m) {
            S.A0 = ldg_half(c.ws, k * NB + j0, m, lane);
            S.A1 = ldg_half(c.ws, a1b ? k * NB + p.j1 : c.zb, m, lane);
            S.B0 = ldg_half(c.ws, (k >= ks0) ? k * NB + c0 : c.zb, m, lane);
            S.B1 = ldg_half(c.ws, (k >= ks1) ? k * NB + c1 : c.zb, m, lane);
        };
        // three operand sets in rotation: the halves of the next TWO half steps are in flight; past the end the last half
        // again, unused (n >= 2 here).  One workgroup alone on a CU spends 1 900-2 700 cycles per half step against ~1 000 of
        // MFMA and split work; with two per CU the other one fills most of that (E57: configs[2] +1.8 %, configs[1] +0.2 %)
        const int n = 2 * (j0 - kmin);
        auto load_i = [&](KinvOps& S, int i) { const int ii = min(i, n - 1); load(S, kmin + (ii >> 1), ii & 1); };
        KinvOps S0, S1, S2;
        load_i(S0, 0);
        load_i(S1, 1);
        int i = 0;
        for (; i + 3 <= n; i += 3) {
            load_i(S2, i + 2);
            kinv_comp<false>(W, S0);
            load_i(S0, i + 3);
            kinv_comp<false>(W, S1);
            load_i(S1, i + 4);
            kinv_comp<false>(W, S2);
        }
        if (i < n) { kinv_comp<false>(W, S0); ++i; }
        if (i < n) kinv_comp<false>(W, S1);
    }
    // U-type: W = K - acc ; M-type: W = -acc
    if (v0 && u0) {
        W[0] = kblock<D, KN>(c, j0, c0) - W[0];
        if (p.has1) W[2] = kblock<D, KN>(c, p.j1, c0) - W[2];
    } else {
        W[0] = -W[0];
        W[2] = -W[2];
    }
    if (v1 && u1) {
        W[1] = kblock<D, KN>(c, j0, c1) - W[1];
        if (p.has1) W[3] = kblock<D, KN>(c, p.j1, c1) - W[3];
    } else {
        W[1] = -W[1];
        W[3] = -W[3];
    }
}

// row r of a group: X = L_jr^-1 W_r, store, alpha update; r = 0 also folds row j0 into the row-j1 RHS.
// `par` selects the LDS copy of the panel's factors (double-buffered across panels, see phase_pt).
// apsum[n]: item n's contribution to alpha (M-type items), summed over the panel's rows and added to alpha ONCE after the
// last row -- one read-modify-write per group and column, whoever runs the group.
// COOP: the panel's factors come from the workspace (where the chain stores the same values it keeps in LDS for its own
// workgroup), alpha lives in the owner's workspace.
template <int D, int KN, bool COOP>
__device__ __forceinline__ void pt_group_row(const Ctx<D, KN>& c, const Panel<D>& p, int g, int r, int par, f32x16 (&W)[4],
                                             float (&apsum)[2]) {
    const int NB = c.NB, lane = c.lane;
    const int jr = p.j0 + r;
    const f32x16 Lop = COOP ? ldg(c.ws, c.dT0 + jr, lane) : ldl(c.L.LT + (2 * par + r) * BLK, lane);
    f32x16 U01 = Lop;
    const bool fold = (r == 0) && p.has1;
    if (fold) U01 = COOP ? ldg(c.ws, p.j0 * NB + p.j1, lane) : ldl(c.L.U01 + par * BLK, lane);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int e = 2 * g + n;
        if (e < p.nItems) {
            const bool isu = e < p.nU;
            const int col = pt_item_col(p, e);
            f32x16 src = W[n];
            if (r == 1) src = W[2 + n];
            f32x16 X = zero16();
            mma_blk(X, Lop, src);
            stg(c.ws, jr * NB + col, lane, X);          // U-type: upper slot (jr,col); M-type: lower slot
            if (!isu) {
                float ap = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) ap = fmaf(X[q], lds_f[c.L.z + 32 * jr + rho(q, c.h)], ap);
                ap = xhalf_sum(ap);
                apsum[n] = (r == 0) ? ap : apsum[n] + ap;
                if (r == p.has1 && c.h == 0) {
                    if (COOP) {
                        // ordered against the other updates of this column by colrow (release after the drain below,
                        // acquire = the sc1 poll of the next group that owns the column)
                        gfloat* a = c.ag + 32 * col + c.g;
                        gst_f(a, gld_f(a) + apsum[n]);
                    } else {
                        lds_f[c.L.alpha + 32 * col + c.g] += apsum[n];
                    }
                }
            }
            if (fold) {
                f32x16 T = zero16();
                mma_blk(T, U01, X);
                W[2 + n] -= T;
            }
        }
    }
}

// spin on a flag of the sweep in the owner's control block (cooperative evaluation); false = failed, unwind
__device__ __forceinline__ bool pt_wait_g(gCoopCtl* ctl, const gint* flag, int v) {
    int spins = 0;
    while (__hip_atomic_load(flag, RLX_AGENT) < v) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 7) == 0 && __hip_atomic_load(&ctl->fail, RLX_AGENT)) return false;
        if (spins > (1 << 21)) {             // never reached by design (seconds); a lost flag must not hang the GPU
            __hip_atomic_store(&ctl->fail, 2, RLX_AGENT);
            COOP_STAT(ctl, 3);
            return false;
        }
    }
    asm volatile("" ::: "memory");           // the data behind the flag is read with sc1 loads only: no invalidate needed
    return true;
}

// spin on a flag of the sweep (see phase_pt); false = the evaluation has failed, unwind
__device__ __forceinline__ bool pt_wait(Shared* sh, const int* flag, int v) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
        if (__hip_atomic_load(&sh->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {           // never reached by design (seconds); a lost flag must not hang the GPU
            __hip_atomic_store(&sh->fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

// owner-local flag (LDS) of a cooperative evaluation: also gives up when a helper has failed the evaluation
__device__ __forceinline__ bool pt_wait_lc(Shared* sh, gCoopCtl* ctl, const int* flag, int v) {
    int spins = 0;
    while (__hip_atomic_load(flag, RLX_WG) < v) {
        if (__hip_atomic_load(&sh->fail, RLX_WG)) return false;
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (((spins & 63) == 0 && __hip_atomic_load(&ctl->fail, RLX_AGENT)) || spins > (1 << 22)) {
            __hip_atomic_store(&sh->fail, 2, RLX_WG);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

// sweep flags that helpers see live in the owner's control block during a cooperative evaluation, in LDS otherwise
#define PTW(field, v) (COOP ? pt_wait_g(c.ctl, &c.ctl->field, (v)) : pt_wait(sh, &sh->field, (v)))
#define PTW_LOCAL(field, v) (COOP ? pt_wait_lc(sh, c.ctl, &sh->field, (v)) : pt_wait(sh, &sh->field, (v)))
#define PTS(field, v)                                                                     \
    do {                                                                                  \
        if (COOP) __hip_atomic_store(&c.ctl->field, (v), RLX_AGENT);                      \
        else __hip_atomic_store(&sh->field, (v), RLX_WG);                                 \
    } while (0)
// Before a flag that announces this wave's stores: the wave waits until they have completed.  Workspace blocks are stored
// write-through (sc1) and loaded past L1 (sc1) in every mode, so that a tile's data is coherent whichever workgroups touch
// it from one evaluation to the next; a write-through store that is still on its way when another wave's load of the same
// block arrives is NOT ordered before that load, even on the same CU (seen as one wrong tile in 200 000 when the flag was
// raised behind a workgroup-scope fence only, which waits for nothing on gfx950).
#define PT_RELEASE() coop_drain()

#ifdef GPSAT_DUMP       // diagnostic build: stages of the forward-solve partial sums per lane (EXPERIMENTS.md E48), rows < 32
#define DBG_TP(stage, jr, v) do { if ((jr) < 32) c.ws[(size_t)(c.zb - 8) * BLK + ((stage) * 32 + (jr)) * 64 + lane] = (v); } while (0)
#else
#define DBG_TP(stage, jr, v) do {} while (0)
#endif

// D00 += U_k,j0^T U_k,j0, D01 += U_k,j0^T U_k,j1, D11 += U_k,j1^T U_k,j1 and the forward-solve partials for k in [kb, ke)
template <int D, int KN>
__device__ __forceinline__ void chain_kloop(const Ctx<D, KN>& c, const Panel<D>& p, int kb, int ke, f32x16& D00, f32x16& D01,
                                            f32x16& D11, float& tp0, float& tp1) {
    const int NB = c.NB, lane = c.lane, j0 = p.j0, j1 = p.j1;
    const bool has1 = p.has1 != 0;
    if (kb >= ke) return;
    // three-plane bf16 products (see pt_group_kloop): per half step 18 MFMAs (D00, D01, D11), the forward-solve sums from the
    // fp32 halves as they were loaded
    auto load = [&](RawHalf& a0, RawHalf& a1, int k, int m) {
        a0 = ldg_half(c.ws, k * NB + j0, m, lane);
        a1 = ldg_half(c.ws, has1 ? k * NB + j1 : c.zb, m, lane);
    };
    auto comp = [&](const RawHalf& a0, const RawHalf& a1, int k, int m) {
        HalfPl P0 = split_half(a0), P1 = split_half(a1);
        GPSAT_PLANES_SETTLE(GPSAT_PL(P0), GPSAT_PL(P1));
#define GPSAT_PP(i, j)                          \
        D00 = mfma_bf(P0.p[i], P0.p[j], D00);   \
        D01 = mfma_bf(P0.p[i], P1.p[j], D01);   \
        D11 = mfma_bf(P1.p[i], P1.p[j], D11);
        GPSAT_PP(1, 1) GPSAT_PP(0, 2) GPSAT_PP(2, 0) GPSAT_PP(0, 1) GPSAT_PP(1, 0) GPSAT_PP(0, 0)
#undef GPSAT_PP
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float zk = lds_f[c.L.z + 32 * k + rho(8 * m + q, c.h)];
            tp0 = fmaf(__uint_as_float(a0.q[q >> 2][q & 3]), zk, tp0);
            tp1 = fmaf(__uint_as_float(a1.q[q >> 2][q & 3]), zk, tp1);
        }
    };
    // four operand sets in rotation: three half steps of loads in flight (two half blocks each)
    const int n = 2 * (ke - kb);
    RawHalf a0[4], a1[4];
    // past the end: the last half again, unused (an empty range reads row kb, which exists)
    auto load_i = [&](int slot, int i) { const int ii = max(min(i, n - 1), 0); load(a0[slot], a1[slot], kb + (ii >> 1), ii & 1); };
    load_i(0, 0); load_i(1, 1); load_i(2, 2);
    int i = 0;
    for (; i + 4 <= n; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_i((u + 3) & 3, i + u + 3);
            comp(a0[u], a1[u], kb + ((i + u) >> 1), (i + u) & 1);
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
        if (i + u < n) comp(a0[u], a1[u], kb + ((i + u) >> 1), (i + u) & 1);
}

// The diagonal chain of panel p (one wave): D00/D01/D11 accumulation over k < j0, the two 32x32
// factorisations, U_j0j1, the forward solve z and M_j1,j0.  Publishes the factors in LDS copy `par`.
// `slot` is the panel's index.  kwait: rows >= kwait of the panel columns (= the rows of the previous panel) arrive
// last: parked as a finished k-loop by the column wave (`held`: the chain completes them itself) or, in the tail, written
// by it (the chain then waits for sh->g0done >= slot).  Returns false when the evaluation has failed (waits unwound).
template <int D, int KN, bool COOP>
__device__ __forceinline__ bool pt_chain(Ctx<D, KN>& c, const Panel<D>& p, const Panel<D>& q, const bool want_m, int par,
                                         int kwait, int slot, bool held, int lt_users) {
    Shared* sh = shared_state();
    const int NB = c.NB, lane = c.lane;
    const int j0 = p.j0, j1 = p.j1;
    const bool has1 = p.has1 != 0;
    f32x16 D00 = zero16(), D01 = zero16(), D11 = zero16();
    float tp0 = 0.f, tp1 = 0.f;
    PROF_BEGIN();
    TRACE(c, 1, slot);
    // rows of the panels <= slot-2 of this panel's two columns (the column wave's group 1 of panel slot-2 was the last)
    if (slot >= 2) {
        if (!PTW(colrow[j0], slot - 1)) return false;
        if (has1 && !PTW(colrow[j1], slot - 1)) return false;
    }
    PROF_END(c, 10);
    // the chain is the critical path of the sweep: let it win VALU / LDS issue arbitration against the
    // MFMA-bound wave of the other resident workgroup that shares this SIMD
    __builtin_amdgcn_s_setprio(3);
    TRACE(c, 2, slot);
    chain_kloop<D, KN>(c, p, 0, kwait, D00, D01, D11, tp0, tp1);
    DBG_TP(0, j0, tp0); DBG_TP(0, j1, tp1);
    TRACE(c, 3, slot);
    if (kwait < j0 && !held) {
        // group 0 of the previous panel is finished by the column wave: wait for its rows
        PROF_END(c, 0);
        if (!PTW(g0done, slot)) { __builtin_amdgcn_s_setprio(0); return false; }
        PROF_END(c, 10);
        chain_kloop<D, KN>(c, p, kwait, j0, D00, D01, D11, tp0, tp1);
    }
    if (kwait < j0 && held) {
        PROF_END(c, 0);
        if (!PTW_LOCAL(parked, slot)) { __builtin_amdgcn_s_setprio(0); return false; }
        PROF_END(c, 10);
        TRACE(c, 4, slot);
        // Rows j0-2, j0-1 of this panel's two columns = group 0 of the previous panel q (both items U-type), whose k-loop the column
        // wave has parked in LDS (L.Wh).  Finish them here (6 block products), store them for
        // everybody else, and use them straight from registers as the last two k-steps of D00 / D01 / D11.
        const int qpar = par ^ 1;
        const f32x16 L0 = ldl(c.L.LT + (2 * qpar + 0) * BLK, lane);
        const f32x16 L1 = ldl(c.L.LT + (2 * qpar + 1) * BLK, lane);
        const f32x16 Uq = ldl(c.L.U01 + qpar * BLK, lane);
        f32x16 X0[2], X1[2];
        f32x16 Wp[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) Wp[n] = ldl(c.L.Wh + n * BLK, lane);
        // the parking area is free again for the column wave (LDS serves a wave's requests in order: the reads above
        // are ahead of this store, and the next parked k-loop is written only after the store has been seen)
        wave_lds_sync();
        if (lane == 0) __hip_atomic_store(&sh->whfree, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f32x16 W0 = Wp[n];
            f32x16 W1 = Wp[2 + n];
            X0[n] = zero16();
            mma_blk(X0[n], L0, W0);
            f32x16 T = zero16();
            mma_blk(T, Uq, X0[n]);
            W1 -= T;
            X1[n] = zero16();
            mma_blk(X1[n], L1, W1);
            stg(c.ws, q.j0 * NB + j0 + n, lane, X0[n]);        // item n of panel q is the U-type column j0 + n
            stg(c.ws, q.j1 * NB + j0 + n, lane, X1[n]);
        }
        mma_blk(D00, X0[0], X0[0]); mma_blk(D00, X1[0], X1[0]);
        mma_blk(D01, X0[0], X0[1]); mma_blk(D01, X1[0], X1[1]);
        mma_blk(D11, X0[1], X0[1]); mma_blk(D11, X1[1], X1[1]);
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
            const float z0 = lds_f[c.L.z + 32 * q.j0 + rho(qq, c.h)], z1 = lds_f[c.L.z + 32 * q.j1 + rho(qq, c.h)];
            tp0 = fmaf(X0[0][qq], z0, fmaf(X1[0][qq], z1, tp0));
            tp1 = fmaf(X0[1][qq], z0, fmaf(X1[1][qq], z1, tp1));
        }
        // the column wave needs these rows for the k-loop it runs ahead at the end of this slot
        PT_RELEASE();
        if (lane == 0) PTS(g0done, slot);
    }
    PROF_END(c, 0);
    DBG_TP(1, j0, tp0); DBG_TP(1, j1, tp1);
    TRACE(c, 5, slot);
    // the factor copies of this parity still serve the groups of panel slot-2 (cooperative evaluation: every group reads
    // the factors from the workspace, the LDS copies serve this chain alone)
    if (slot >= 2 && !COOP) {
        if (!pt_wait(sh, &sh->gdone[par], lt_users)) { __builtin_amdgcn_s_setprio(0); return false; }
        if (lane == 0) __hip_atomic_store(&sh->gdone[par], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PROF_END(c, 10);
    }
    TRACE(c, 6, slot);
    f32x16 S1keep = zero16();
    const int nrow = has1 ? 2 : 1;
    for (int r = 0; r < nrow; ++r) {
        const int jr = j0 + r;
        f32x16 Dd;
        float tp;
        if (r == 0) {
            Dd = kblock<D, KN>(c, j0, j0) - D00;
            tp = tp0;
        } else {
            // D11 <- K_j1j1 - sum_{k<j0} .. - U01^T U01 ; t1 += U01^T z_j0
            const f32x16 U01 = ldl(c.L.U01 + par * BLK, lane);
            mma_blk(D11, U01, U01);
#pragma unroll
            for (int q = 0; q < 16; ++q) tp1 = fmaf(U01[q], lds_f[c.L.z + 32 * j0 + rho(q, c.h)], tp1);
            Dd = kblock<D, KN>(c, j1, j1) - D11;
            tp = tp1;
        }
        f32x16 S1, S2;
        double ls;
        int bad;
        PROF_END(c, 2);
        diag_factor(Dd, c.L.Ad, c.L.piv, lane, S1, S2, ls, bad);
        PROF_END(c, 1);
        TRACE(c, 7 + r, slot);
        stg(c.ws, jr * NB + jr, lane, S1);            // M_jrjr
        stg(c.ws, c.dT0 + jr, lane, S2);              // (L_jr^-1)^T
        stl(c.L.LT + (2 * par + r) * BLK, lane, S2);
        // z_jr = X (y_jr - t_jr): lane (h,g) holds X[g][rho(r,h)] in S2
        const float t = xhalf_sum(tp);
        DBG_TP(2, jr, tp); DBG_TP(3, jr, t);
        if (c.h == 0) lds_f[c.L.tmp + c.g] = lds_f[c.L.y + 32 * jr + c.g] - t;
        wave_lds_sync();
        float zz = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) zz = fmaf(S2[q], lds_f[c.L.tmp + rho(q, c.h)], zz);
        zz = xhalf_sum(zz);
        if (c.h == 0) {
            lds_f[c.L.z + 32 * jr + c.g] = zz;
            if (COOP) gst_f(c.zg + 32 * jr + c.g, zz);
        }
        if (lane == 0) {
            sh->logdet += ls;
            if (bad) { sh->fail = 1; if (COOP) { __hip_atomic_store(&c.ctl->fail, 1, RLX_AGENT); COOP_STAT(c.ctl, 5); } }
        }
        wave_lds_sync();
        if (want_m) {
            // alpha_jr += M_jrjr^T z_jr
            float ap = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) ap = fmaf(S1[q], lds_f[c.L.z + 32 * jr + rho(q, c.h)], ap);
            ap = xhalf_sum(ap);
            if (c.h == 0) {
                if (COOP) { gfloat* a = c.ag + 32 * jr + c.g; gst_f(a, gld_f(a) + ap); }
                else lds_f[c.L.alpha + 32 * jr + c.g] += ap;
            }
        }
        if (r == 0) {
            S1keep = S1;
            if (has1) {
                D01 = kblock<D, KN>(c, j0, j1) - D01;
                f32x16 U01 = zero16();
                mma_blk(U01, S2, D01);
                stg(c.ws, j0 * NB + j1, lane, U01);
                stl(c.L.U01 + par * BLK, lane, U01);
                wave_lds_sync();
            }
        } else if (want_m) {
            // M_j1,j0 = -L_j1^-1 U01^T M_j0j0
            const f32x16 U01 = ldl(c.L.U01 + par * BLK, lane);
            f32x16 T = zero16();
            mma_blk(T, U01, S1keep);
            T = -T;
            f32x16 Mx = zero16();
            mma_blk(Mx, S2, T);
            stg(c.ws, j1 * NB + j0, lane, Mx);
            float ap = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) ap = fmaf(Mx[q], lds_f[c.L.z + 32 * j1 + rho(q, c.h)], ap);
            ap = xhalf_sum(ap);
            if (c.h == 0) {
                if (COOP) { gfloat* a = c.ag + 32 * j0 + c.g; gst_f(a, gld_f(a) + ap); }
                else lds_f[c.L.alpha + 32 * j0 + c.g] += ap;
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    // publish: factors (LDS + workspace), diagonal blocks, U_j0j1, M_j1j0, z are in place; the panel's columns are complete
    PT_RELEASE();
    if (lane == 0) {
        PTS(colrow[j0], slot + 1);
        if (has1) PTS(colrow[j1], slot + 1);
        PTS(ready, slot + 1);
    }
    PROF_END(c, 2);
    TRACE(c, 9, slot);
    return true;
}

template <int D>
__device__ __forceinline__ Panel<D> make_panel(int NB, int pi, bool want_m) {
    Panel<D> p;
    p.j0 = 2 * pi;
    p.j1 = p.j0 + 1;
    p.has1 = (p.j1 < NB) ? 1 : 0;
    p.nU = p.has1 ? (NB - 1 - p.j1) : 0;
    p.nItems = p.nU + (want_m ? p.j0 : 0);
    return p;
}

// Dataflow sweep: no workgroup barrier between panels; the waves meet through flags in LDS (Shared):
//   ready      panels whose diagonal chain is complete (factors published, diagonal blocks / U_j0j1 / M_j1j0 / z in place)
//   colrow[c]  panels whose rows of block column c are in memory (a group of panel s needs colrow >= s for its columns)
//   parked / whfree   hand-over of the run-ahead k-loop of group 0 (L.Wh) from the column wave to the chain and back
//   g0done     rows of panel s-1 of the columns of panel s are in memory (chain -> column wave, or the reverse in the tail)
//   gdone[par] groups of the panel of that parity that have finished with its factor copies (the chain of panel s+2 reuses them)
//   wave 0 ("chain")  : the diagonal chains of all panels back to back, each as soon as its two columns are there;
//   wave 1 ("column") : per panel s the work the next chains wait for -- group 0 (tail only) and group 1 of panel s-1, then
//                       the k-loop of group 0 of panel s run ahead and parked for the chain;
//   other waves ("bulk"): groups >= 2 of all panels from ONE queue in panel-major order; waves 0 and 1 join when done.
// The dependent path per panel is: 6 block products (rows of group 0) -> 2 k-steps + two 32x32 factorisations; nobody
// waits for anything else, so the second workgroup of the CU fills whatever a wave leaves idle.
// Results do not depend on which wave runs a group: each group is the same arithmetic and per-column updates are ordered
// by colrow.
// groups of panel q that go through pt_run_group (group 0 is finished by the next chain when it was parked)
template <int D>
__device__ __forceinline__ int pt_lt_users(const Panel<D>& q) { return ((q.nItems + 1) >> 1) - (q.nU >= 2 ? 1 : 0); }

// one group of panel q (index sq): wait for its columns, k-loop, wait for the panel's factors, row solves, publish.
// `queued`: the group came from the bulk queue (cooperative evaluation: counted in ctl->done).
template <int D, int KN, bool COOP>
__device__ __forceinline__ bool pt_run_group(const Ctx<D, KN>& c, const Panel<D>& q, int sq, int g, bool queued) {
    Shared* sh = shared_state();
    PROF_BEGIN();
    TRACE(c, 20, sq * 16 + g);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int e = 2 * g + n;
        if (e < q.nItems && !PTW(colrow[pt_item_col(q, e)], sq)) return false;
    }
    // ... and of the panel's own two columns (their rows of panel sq-1 come last: group 0 of that panel)
    if (sq >= 1 && !PTW(g0done, sq)) return false;
    PROF_END(c, 3);
    TRACE(c, 21, sq * 16 + g);
    f32x16 W[4];
    pt_group_kloop<D, KN>(c, q, g, W);
    PROF_END(c, 4);
    TRACE(c, 22, sq * 16 + g);
    // only the row solves need the panel's own factors: the k-loop above ran while its chain may still be at work
    if (!PTW(ready, sq + 1)) return false;
    if (COOP && c.helper && 2 * g + 1 >= q.nU) {
        // a helper has no z of its own: the panel's two rows of z (written by the owner's chain before `ready`) come from the
        // owner's workspace into this workgroup's LDS (several waves may copy the same values at once)
        const int i = 32 * q.j0 + c.lane;
        if (i < c.Npad) lds_f[c.L.z + i] = gld_f(c.zg + i);
        wave_lds_sync();
    }
    PROF_END(c, 3);
    TRACE(c, 24, sq * 16 + g);
    float apsum[2] = {0.f, 0.f};
    pt_group_row<D, KN, COOP>(c, q, g, 0, sq & 1, W, apsum);
    if (q.has1) pt_group_row<D, KN, COOP>(c, q, g, 1, sq & 1, W, apsum);
    PT_RELEASE();
    if (c.lane == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int e = 2 * g + n;
            if (e < q.nItems) PTS(colrow[pt_item_col(q, e)], sq + 1);
        }
        if (COOP) { if (queued) __hip_atomic_fetch_add(&c.ctl->done, 1, RLX_AGENT); if (c.helper) COOP_STAT(c.ctl, 2); }
        else __hip_atomic_fetch_add(&sh->gdone[sq & 1], 1, RLX_WG);
    }
    PROF_END(c, 5);
    TRACE(c, 23, sq * 16 + g);
    return true;
}

// groups of the bulk queue: groups >= 2 of every panel, panel-major; nobody waits for the last panel's groups 0 and 1, so
// they are queued as well
template <int D>
__device__ __forceinline__ int pt_queue_len(int NB, bool want_m) {
    const int NP = (NB + 1) >> 1;
    int tot = 0;
    for (int s = 0; s < NP; ++s) {
        const Panel<D> q = make_panel<D>(NB, s, want_m);
        tot += max(0, ((q.nItems + 1) >> 1) - ((s == NP - 1) ? 0 : 2));
    }
    return tot;
}

// the bulk loop of one wave: pull group indices until the queue is exhausted (owner and helpers alike)
template <int D, int KN, bool COOP>
__device__ __forceinline__ bool pt_bulk_loop(const Ctx<D, KN>& c, const bool want_m, bool ok) {
    Shared* sh = shared_state();
    const int NB = c.NB;
    const int NP = (NB + 1) >> 1;
    int sq = 0, base = 0;
    Panel<D> q = make_panel<D>(NB, 0, want_m);
    int g0 = (NP == 1) ? 0 : 2;
    int nq = max(0, ((q.nItems + 1) >> 1) - g0);
    while (ok) {
        int idx;
        if (COOP) {
            int v = 0;
            if (c.lane == 0) v = __hip_atomic_fetch_add(&c.ctl->qhead, 1, RLX_AGENT);
            idx = __builtin_amdgcn_readfirstlane(v);
        } else {
            idx = wave_pull(&sh->qhead, c.lane);
        }
        while (sq < NP && idx >= base + nq) {
            base += nq;
            ++sq;
            if (sq < NP) {
                q = make_panel<D>(NB, sq, want_m);
                g0 = (sq == NP - 1) ? 0 : 2;
                nq = max(0, ((q.nItems + 1) >> 1) - g0);
            }
        }
        if (sq >= NP) break;
        TRACE(c, 50, sq);
        ok = pt_run_group<D, KN, COOP>(c, q, sq, g0 + idx - base, true);
    }
    return ok;
}

// bounded wait of the owner's thread 0 for a counter of its control block
__device__ __forceinline__ bool coop_wait_eq(gCoopCtl* ctl, const gint* word, int v) {
    for (int spins = 0; __hip_atomic_load(word, RLX_AGENT) != v; ++spins) {
        __builtin_amdgcn_s_sleep(8);
        if (spins > (1 << 21)) { COOP_STAT(ctl, 4); return false; }
        if ((spins & 15) == 15 && __hip_atomic_load(&ctl->fail, RLX_AGENT)) return false;
    }
    return true;
}

// owner, thread 0: open a phase of the cooperative evaluation (everything the helpers read has been stored and drained)
__device__ __forceinline__ void coop_open(Shared* sh, gCoopCtl* ctl, int kind) {
    sh->coop_seq += 1;
    __hip_atomic_store(&ctl->pa, PA_MAKE(sh->coop_seq, kind), RLX_AGENT);        // nobody is checked in: the last phase was closed
}

// owner, thread 0: all `total` queue groups done (or the evaluation failed), then close the phase and wait until the helpers
// that checked in have checked out.  Returns false when a wait gave up.
__device__ __forceinline__ bool coop_close(Shared* sh, gCoopCtl* ctl, int total) {
    bool ok = coop_wait_eq(ctl, &ctl->done, total);
    // kind -> CLOSED (0) with the count untouched, then the count of the same word (a helper's compare-and-swap either came
    // before this AND and is counted, or fails on the changed phase half)
    __hip_atomic_fetch_and(&ctl->pa, ~(3ull << 32), RLX_AGENT);
    for (int spins = 0; PA_ACTIVE(__hip_atomic_load(&ctl->pa, RLX_AGENT)) != 0u; ++spins) {
        __builtin_amdgcn_s_sleep(8);
        if (spins > (1 << 21)) { COOP_STAT(ctl, 4); ok = false; break; }     // never reached by design: a lost helper must not hang the GPU
    }
    return ok;
}

template <int D, int KN, bool COOP>
__device__ __forceinline__ void phase_pt(Ctx<D, KN>& c, const bool want_m) {
    Shared* sh = shared_state();
    const int NB = c.NB, w = c.w;
    const int NP = (NB + 1) >> 1;
    if (c.tid == 0) {
        sh->logdet = 0.0; sh->fail = 0; sh->g0done = 0; sh->ready = 0; sh->parked = 0; sh->whfree = 0;
        sh->gdone[0] = 0; sh->gdone[1] = 0; sh->qhead = 0;
    }
    for (int idx = c.tid; idx < NB; idx += NT) sh->colrow[idx] = 0;
    for (int idx = c.tid; idx < c.Npad; idx += NT) lds_f[c.L.alpha + idx] = 0.f;
    if (COOP) {
        // the flags, counters and alpha of this evaluation in the control block / workspace, then the phase word: helpers
        // touch none of them before they have seen the new phase, and no helper of an earlier phase is left (coop_close)
        if (c.tid == 0) {
            __hip_atomic_store(&c.ctl->ready, 0, RLX_AGENT);
            __hip_atomic_store(&c.ctl->g0done, 0, RLX_AGENT);
            __hip_atomic_exchange(&c.ctl->qhead, 0, RLX_AGENT);
            __hip_atomic_exchange(&c.ctl->done, 0, RLX_AGENT);
            __hip_atomic_exchange(&c.ctl->fail, 0, RLX_AGENT);
            __hip_atomic_store(&c.ctl->want_m, want_m ? 1 : 0, RLX_AGENT);
        }
        for (int idx = c.tid; idx < NB; idx += NT) __hip_atomic_store(&c.ctl->colrow[idx], 0, RLX_AGENT);
        for (int idx = c.tid; idx < c.Npad; idx += NT) gst_f(c.ag + idx, 0.f);
        coop_drain();
        __syncthreads();
        if (c.tid == 0) { coop_open(sh, c.ctl, COOP_SWEEP); COOP_STAT(c.ctl, 0); }
    }
    __syncthreads();
    bool ok = true;
    if (w == 0) {
        for (int s = 0; s < NP && ok; ++s) {
            const Panel<D> p = make_panel<D>(NB, s, want_m);
            const Panel<D> q = make_panel<D>(NB, s > 0 ? s - 1 : 0, want_m);
            const int kwait = (s > 0) ? (p.j0 - 2) : p.j0;
            const bool held = (s > 0) && (q.nU >= 2);
            const int users = (s >= 2) ? pt_lt_users<D>(make_panel<D>(NB, s - 2, want_m)) : 0;
            ok = pt_chain<D, KN, COOP>(c, p, q, want_m, s & 1, kwait, s, held, users);
            if (ok && __hip_atomic_load(&sh->fail, RLX_WG)) ok = false;
        }
    } else if (w == 1) {
        for (int s = 0; s < NP && ok; ++s) {
            if (s >= 1) {
                const Panel<D> q = make_panel<D>(NB, s - 1, want_m);
                const int nGroups = (q.nItems + 1) >> 1;
                if (ok && nGroups > 0 && q.nU < 2) {
                    // tail: group 0 was not parked; the chain of panel s waits for these rows
                    ok = pt_run_group<D, KN, COOP>(c, q, s - 1, 0, false);
                    if (ok && c.lane == 0) PTS(g0done, s);
                }
                if (NW < 8 && ok && nGroups > 1) ok = pt_run_group<D, KN, COOP>(c, q, s - 1, 1, false);
            }
            if (ok && s < NP) {
                // run ahead: k-loop of group 0 of panel s (the columns of the next chain), parked in LDS for that chain
                const Panel<D> pn = make_panel<D>(NB, s, want_m);
                if (pn.nU >= 2) {
                    PROF_BEGIN();
                    ok = PTW(colrow[pn.j1 + 1], s) && PTW(colrow[pn.j1 + 2], s);
                    // rows of panel s-1 of the columns j0(s), j1(s): written by the chain of panel s (or above, in the tail)
                    if (ok && s >= 1) ok = PTW(g0done, s);
                    if (ok) ok = PTW_LOCAL(whfree, s);
                    PROF_END(c, 3);
                    TRACE(c, 30, s);
                    if (ok) {
                        f32x16 W[4];
                        pt_group_kloop<D, KN>(c, pn, 0, W);
#pragma unroll
                        for (int n = 0; n < 4; ++n) stl(c.L.Wh + n * BLK, c.lane, W[n]);
                        wave_lds_sync();
                        if (c.lane == 0) __hip_atomic_store(&sh->parked, s + 1, RLX_WG);
                        PROF_END(c, 4);
                        TRACE(c, 31, s);
                    }
                }
            }
        }
    }
    if (NW >= 8 && w == 2) {
        // 8-wave build: group 1 of every panel (the columns of the chain after next) has a wave of its own, so that the
        // column wave's run-ahead k-loop of panel s and group 1 of panel s-1 -- both on the path the chains wait for, both
        // growing with the panel index -- run side by side (one wave doing both bounded a helped 2048-point tile)
        for (int s = 1; s < NP && ok; ++s) {
            const Panel<D> q = make_panel<D>(NB, s - 1, want_m);
            if (((q.nItems + 1) >> 1) > 1) ok = pt_run_group<D, KN, COOP>(c, q, s - 1, 1, false);
        }
    }
    ok = pt_bulk_loop<D, KN, COOP>(c, want_m, ok);
    if (COOP && !ok && c.lane == 0) {
        // a wave of the owner unwinds: everybody else must, too
        __hip_atomic_store(&sh->fail, 2, RLX_WG);
        __hip_atomic_store(&c.ctl->fail, 2, RLX_AGENT);
    }
    TRACE(c, 60, 0);
    coop_drain();                     // the blocks this wave stored are read by other waves behind the barrier
    __syncthreads();
    if (COOP) {
        // the helpers' groups are part of this sweep: all queue groups done, phase closed, helpers out; then alpha (and a
        // failure a helper met) come home
        if (c.tid == 0) {
            if (!coop_close(sh, c.ctl, pt_queue_len<D>(NB, want_m)) && !sh->fail) sh->fail = 2;
            if (__hip_atomic_load(&c.ctl->fail, RLX_AGENT) && !sh->fail) sh->fail = 2;
        }
        __syncthreads();
        if (want_m && !sh->fail)
            for (int idx = c.tid; idx < c.Npad; idx += NT) lds_f[c.L.alpha + idx] = gld_f(c.ag + idx);
        __syncthreads();
    }
    TRACE(c, 61, 0);
}

// ---------------------------------------------------------------------------------------------
// phase G: (K^-1)_ab = sum_{c>=a} M_ca^T M_cb for a >= b, by groups of 2 block rows x 2 block columns
// per wave (4 accumulators, double-buffered operand registers), contracted in registers against
// dK/dtheta recomputed on the fly (K^-1 is never stored).  Writes sh->gth (dNLL/dtheta).
// Groups are pulled from ONE queue, largest first (group index = ia (ia + 1) / 2 + ib for the block-row pair ia and the
// block-column pair ib <= ia; its k-loop has NB - 2 ia steps).  Every group leaves its D + 2 partial sums PER LANE in the
// workspace (gpart, aliasing the prediction scratch, which is idle during an evaluation); they are then added up in a
// fixed order -- the gradient does not depend on which wave ran which group (bit for bit), so the queue may be dynamic.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_part(float* __restrict__ ws, int byte_off, int lane, float v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, lane * 4, byte_off, GPSAT_ST_AUX);
}

__device__ __forceinline__ float ld_part(const float* __restrict__ ws, int byte_off, int lane) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, 0x7fffffff, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, byte_off, GPSAT_LD_AUX));
}

// (KinvOps / kinv_comp: next to mma_half above)
template <bool DIAG>
__device__ __forceinline__ void kinv_load(KinvOps& S, const float* __restrict__ ws, int NB, int cc, int m, int a0, int b0, int lane) {
    S.A0 = ldg_half(ws, cc * NB + a0, m, lane);
    S.A1 = ldg_half(ws, cc * NB + a0 + 1, m, lane);
    if (!DIAG) {
        S.B0 = ldg_half(ws, cc * NB + b0, m, lane);
        S.B1 = ldg_half(ws, cc * NB + b0 + 1, m, lane);
    }
}

// acc[0..3] = the blocks (a0, b0), (a0, b0 + 1), (a0 + 1, b0), (a0 + 1, b0 + 1) of K^-1 = M^T M (what a missing block row or
// column leaves in its accumulators is not used by the caller)
template <bool DIAG>
__device__ __forceinline__ void kinv_kloop(const float* __restrict__ ws, int NB, int a0, int b0, int lane, f32x16 (&acc)[4]) {
    const int a1 = a0 + 1;
    // block row a0: only the A0 products (M_a0,a1 = 0)
    RawHalf fA[2], fB0[2], fB1[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        fA[m] = ldg_half(ws, a0 * NB + a0, m, lane);
        if (!DIAG) { fB0[m] = ldg_half(ws, a0 * NB + b0, m, lane); fB1[m] = ldg_half(ws, a0 * NB + b0 + 1, m, lane); }
    }
    // 8-wave build: three operand sets in rotation, two half steps of loads in flight (configs[2] +0.4 %); in the 4-wave build
    // the third set spills (configs[1] -0.9 %) and the CU's other workgroup covers the wait anyway: two sets in ping-pong
#ifdef GPSAT_W8
    const int n = 2 * (NB - a1);
    auto load_i = [&](KinvOps& S, int i) { const int ii = max(min(i, n - 1), 0); kinv_load<DIAG>(S, ws, NB, min(a1 + (ii >> 1), NB - 1), ii & 1, a0, b0, lane); };
    KinvOps S0, S1, S2;
    if (n > 0) { load_i(S0, 0); load_i(S1, 1); }
#else
    KinvOps S0, S1;
    if (a1 < NB) kinv_load<DIAG>(S0, ws, NB, a1, 0, a0, b0, lane);
#endif
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        HalfPl A = split_half(fA[m]);
        if (DIAG) { GPSAT_PLANES_SETTLE(GPSAT_PL(A)); mma_half(acc[0], A, A); }
        else {
            HalfPl B0 = split_half(fB0[m]), B1 = split_half(fB1[m]);
            GPSAT_PLANES_SETTLE(GPSAT_PL(A), GPSAT_PL(B0), GPSAT_PL(B1));
            mma_half(acc[0], A, B0);
            mma_half(acc[1], A, B1);
        }
    }
#ifdef GPSAT_W8
    int i = 0;
    for (; i + 3 <= n; i += 3) {
        load_i(S2, i + 2);
        kinv_comp<DIAG>(acc, S0);
        load_i(S0, i + 3);
        kinv_comp<DIAG>(acc, S1);
        load_i(S1, i + 4);
        kinv_comp<DIAG>(acc, S2);
    }
    if (i < n) { kinv_comp<DIAG>(acc, S0); ++i; }
    if (i < n) kinv_comp<DIAG>(acc, S1);
#else
    for (int cc = a1; cc < NB; ++cc) {
        kinv_load<DIAG>(S1, ws, NB, cc, 1, a0, b0, lane);
        kinv_comp<DIAG>(acc, S0);
        kinv_load<DIAG>(S0, ws, NB, min(cc + 1, NB - 1), 0, a0, b0, lane);     // past the end: the last row again, unused
        kinv_comp<DIAG>(acc, S1);
    }
#endif
}

// one K^-1 group (a0 = 2 ia, b0 = 2 ib): k-loop, contraction, per-lane partial sums -> gpart[g]
template <int D, int KN>
__device__ __forceinline__ void grad_group(const Ctx<D, KN>& c, int g, int ia, int ib) {
    const int NB = c.NB, lane = c.lane;
    const int a0 = 2 * ia, b0 = 2 * ib;
    const int a1 = a0 + 1;
    const bool hasa1 = a1 < NB;
    const int bmax = hasa1 ? a1 : a0;
    PROF_BEGIN();
    const int b1 = b0 + 1;
    const bool hasb1 = b1 <= bmax;
    TRACE(c, 70, a0);
    const bool use01 = hasb1 && b1 <= a0;      // (a0, b1) is a lower block (false on the diagonal group)
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = zero16();
    if (ia == ib) kinv_kloop<true>(c.ws, NB, a0, b0, lane, acc);
    else kinv_kloop<false>(c.ws, NB, a0, b0, lane, acc);
    PROF_END(c, 6);
    float accl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) accl[d] = 0.f;
    float accsf = 0.f, accsn = 0.f;
    contract<D, KN>(c, acc[0], a0, b0, (a0 == b0) ? 1.f : 2.f, accl, accsf, accsn);
    if (use01) contract<D, KN>(c, acc[1], a0, b1, (a0 == b1) ? 1.f : 2.f, accl, accsf, accsn);
    if (hasa1) {
        contract<D, KN>(c, acc[2], a1, b0, (a1 == b0) ? 1.f : 2.f, accl, accsf, accsn);
        if (hasb1) contract<D, KN>(c, acc[3], a1, b1, (a1 == b1) ? 1.f : 2.f, accl, accsf, accsn);
    }
    const int base = c.gp0 + g * ((D + 2) * 256);          // bytes: [D + 2][64 lanes] floats per group
#pragma unroll
    for (int d = 0; d < D; ++d) st_part(c.ws, base + d * 256, lane, accl[d]);
    st_part(c.ws, base + D * 256, lane, accsf);
    st_part(c.ws, base + (D + 1) * 256, lane, accsn);
    PROF_END(c, 7);
    TRACE(c, 71, a0);
}

// the group loop of one wave (owner and helpers alike)
template <int D, int KN, bool COOP>
__device__ __forceinline__ void grad_loop(const Ctx<D, KN>& c) {
    Shared* sh = shared_state();
    const int NBp = (c.NB + 1) >> 1;
    const int ngroups = NBp * (NBp + 1) / 2;
    int ia = 0;
    for (;;) {
        int g;
        if (COOP) {
            int v = 0;
            if (c.lane == 0) v = __hip_atomic_fetch_add(&c.ctl->qhead, 1, RLX_AGENT);
            g = __builtin_amdgcn_readfirstlane(v);
        } else {
            g = wave_pull(&sh->gradnext, c.lane);                    // zeroed by finish_nll
        }
        if (g >= ngroups) break;
        while ((ia + 1) * (ia + 2) / 2 <= g) ++ia;
        grad_group<D, KN>(c, g, ia, g - ia * (ia + 1) / 2);
        if (COOP) {
            coop_drain();
            if (c.lane == 0) __hip_atomic_fetch_add(&c.ctl->done, 1, RLX_AGENT);
        }
    }
}

template <int D, int KN, bool COOP>
__device__ __forceinline__ void phase_grad(Ctx<D, KN>& c) {
    Shared* sh = shared_state();
    const int NB = c.NB, lane = c.lane;
    const int NBp = (NB + 1) >> 1;
    const int ngroups = NBp * (NBp + 1) / 2;
    if (COOP) {
        if (c.tid == 0) {
            __hip_atomic_exchange(&c.ctl->qhead, 0, RLX_AGENT);
            __hip_atomic_exchange(&c.ctl->done, 0, RLX_AGENT);
            coop_drain();
            coop_open(sh, c.ctl, COOP_GRAD);
        }
        __syncthreads();
    }
    grad_loop<D, KN, COOP>(c);
    // the partial sums of all groups are in memory: every wave's stores have completed before the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        PROF_BEGIN();
        __syncthreads();
        PROF_END(c, 8);
    }
    if (COOP) {
        if (c.tid == 0 && !coop_close(sh, c.ctl, ngroups)) sh->fail = 2;
        __syncthreads();
    }
    // fixed-order sum over eight VIRTUAL waves, whatever the build (the 4-wave build's waves run two each): virtual wave vw adds
    // the groups vw, vw + 8, ... per lane (fp64), then across lanes; thread 0 adds the eight in order.  The 4-wave and the
    // 8-wave build thus return the same bits (tests/test_gpu_builds_agree.py): which build runs a tile depends on the batch.
#pragma unroll
    for (int vv = 0; vv < 8 / NW; ++vv) {
        const int vw = c.w + NW * vv;
        double v[D + 2];
#pragma unroll
        for (int i = 0; i < D + 2; ++i) v[i] = 0.0;
        for (int g = vw; g < ngroups; g += 8) {
            const int base = c.gp0 + g * ((D + 2) * 256);
            float f[D + 2];
#pragma unroll
            for (int i = 0; i < D + 2; ++i) f[i] = ld_part(c.ws, base + i * 256, lane);
#pragma unroll
            for (int i = 0; i < D + 2; ++i) v[i] += (double)f[i];
        }
#pragma unroll
        for (int i = 0; i < D + 2; ++i) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < D + 2; ++i) sh->red[vw][i] = v[i];
        }
    }
    __syncthreads();
    if (c.tid == 0) {
        for (int i = 0; i < D + 2; ++i) {
            double s = 0.0;
            for (int ww = 0; ww < 8; ++ww) s += sh->red[ww][i];
            // scaled diff^2 already carries 1/l^2 (dk/dl = g diff^2 / l^3); kf, g are without sf2
            if (i < D) sh->gth[i] = 0.5 * (double)c.sf2 * (s / (double)KScale<KN>::c2) / sh->theta[i];
            else sh->gth[i] = 0.5 * s;
        }
    }
    __syncthreads();
}

// quadratic form + assemble NLL (all threads)
template <int D, int KN>
__device__ __forceinline__ void finish_nll(Ctx<D, KN>& c) {
    Shared* sh = shared_state();
    // eight virtual waves of 64 lanes, whatever the build (see phase_grad)
#pragma unroll
    for (int vv = 0; vv < 8 / NW; ++vv) {
        const int vw = c.w + NW * vv;
        double q = 0.0;
        for (int p = 64 * vw + c.lane; p < c.N; p += 512) { const double zz = (double)lds_f[c.L.z + p]; q += zz * zz; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
        if (c.lane == 0) sh->red[vw][7] = q;
    }
    __syncthreads();
    if (c.tid == 0) {
        double s = 0.0;
        for (int ww = 0; ww < 8; ++ww) s += sh->red[ww][7];
        sh->nll = 0.5 * s + sh->logdet + 0.5 * (double)c.N * 1.8378770664093453;   // log(2 pi)
        sh->gradnext = 0;                  // group queue of the gradient phase
    }
    __syncthreads();
}

// one objective (+ gradient) evaluation at sh->theta.  On return sh->nll, sh->gth, sh->fail are set.
template <int D, int KN, bool COOP>
__device__ __forceinline__ void evaluate(Ctx<D, KN>& c, bool want_grad) {
    Shared* sh = shared_state();
    __syncthreads();
    PROF_BEGIN();
    float invl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) invl[d] = (float)((double)KScale<KN>::c / sh->theta[d]);
    c.sf2 = (float)sh->theta[D];
    c.sn2 = (float)sh->theta[D + 1];
    for (int idx = c.tid; idx < c.Npad; idx += NT) {
#pragma unroll
        for (int d = 0; d < D; ++d) lds_f[c.L.xsc + d * c.Npad + idx] = lds_f[c.L.xs + d * c.Npad + idx] * invl[d];
    }
    __syncthreads();
    // (Experiment E17: not joining the sweep before the gradient phase -- K^-1 groups started per wave as soon as their
    // block columns were final -- removed the idle tail of the sweep and changed nothing: the other workgroup of the CU
    // already fills it.)
    if (COOP) {
        // the parameters of this evaluation for the helpers (drained with the flags in phase_pt, before the phase opens)
        if (c.tid < D + 2) gst_d(&c.ctl->theta[c.tid], sh->theta[c.tid]);
    }
    phase_pt<D, KN, COOP>(c, want_grad);
    if (sh->fail) {
        if (c.tid == 0) { sh->nll = __builtin_inf(); for (int i = 0; i < D + 2; ++i) sh->gth[i] = 0.0; }
        __syncthreads();
        return;
    }
    finish_nll<D, KN>(c);
    if (want_grad) phase_grad<D, KN, COOP>(c);
    if (COOP && sh->fail) {          // a wait of the gradient phase gave up
        if (c.tid == 0) { sh->nll = __builtin_inf(); for (int i = 0; i < D + 2; ++i) sh->gth[i] = 0.0; }
        __syncthreads();
        return;
    }
    if (c.tid == 0) {
        sh->n_eval += 1;
        if (!(sh->nll == sh->nll)) sh->fail = 1;
    }
    __syncthreads();
    PROF_END(c, 9);
}

// ---------------------------------------------------------------------------------------------
// prediction: V = L^-1 K_* by pairs of 32-column chunks per wave (2 accumulators, double-buffered
// operands), f* = V^T z, f*_var = sf2 - colsum(V^2), y_var = f*_var + sn2
// (GPSat/models/gpflow_models.py:229-243).  Requires U, DinvT, z of a successful phase_pt at the final
// parameters.
// ---------------------------------------------------------------------------------------------
template <int D, int KN>
__device__ __forceinline__ void predict_tile(Ctx<D, KN>& c, const float* __restrict__ Xs, float* __restrict__ fm,
                                             float* __restrict__ fv, float* __restrict__ yv, const float (&invl)[D],
                                             float* __restrict__ fcov) {
    const int NB = c.NB, lane = c.lane;
    const int PC = (c.P + 31) / 32;
    for (int pc = 2 * c.w; pc < PC; pc += 2 * NW) {
        // V = L^-1 K_* of the two chunks: this wave's scratch [2][NB] blocks, or (full covariance wanted) the
        // per-tile store of all chunks
        const int v0 = fcov ? c.cv0 + pc * NB : c.vs0 + c.w * 2 * NB;
        const int qa = 32 * pc + c.g, qb = qa + 32;
        const bool va = qa < c.P, vb = qb < c.P;
        float xa[D], xb[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            xa[d] = va ? Xs[(size_t)qa * D + d] * invl[d] : 0.f;
            xb[d] = vb ? Xs[(size_t)qb * D + d] * invl[d] : 0.f;
        }
        float vsa = 0.f, msa = 0.f, vsb = 0.f, msb = 0.f;
        for (int j = 0; j < NB; ++j) {
            f32x16 acc0 = zero16(), acc1 = zero16();
            if (j > 0) {
                // three-plane bf16 products (see pt_group_kloop): 12 MFMAs per half step
                struct PredOps { RawHalf A, B0, B1; };
                auto load = [&](PredOps& S, int k, int m) {
                    S.A = ldg_half(c.ws, k * NB + j, m, lane);
                    S.B0 = ldg_half(c.ws, v0 + k, m, lane);
                    S.B1 = ldg_half(c.ws, v0 + NB + k, m, lane);
                };
                auto comp = [&](const PredOps& S) {
                    HalfPl A = split_half(S.A), B0 = split_half(S.B0), B1 = split_half(S.B1);
                    GPSAT_PLANES_SETTLE(GPSAT_PL(A), GPSAT_PL(B0), GPSAT_PL(B1));
#define GPSAT_PP(i, jj)                                 \
                    acc0 = mfma_bf(A.p[i], B0.p[jj], acc0); \
                    acc1 = mfma_bf(A.p[i], B1.p[jj], acc1);
                    GPSAT_PP(1, 1) GPSAT_PP(0, 2) GPSAT_PP(2, 0) GPSAT_PP(0, 1) GPSAT_PP(1, 0) GPSAT_PP(0, 0)
#undef GPSAT_PP
                };
                PredOps S0, S1;
                load(S0, 0, 0);
                for (int k = 0; k < j; ++k) {
                    load(S1, k, 1);
                    comp(S0);
                    load(S0, min(k + 1, j - 1), 0);            // past the end: the last row again, unused
                    comp(S1);
                }
            }
            const f32x16 Lop = ldg(c.ws, c.dT0 + j, lane);
            f32x16 Wa = ksblock<D, KN>(c, j, xa, va) - acc0;
            f32x16 Va = zero16();
            mma_blk(Va, Lop, Wa);
            stg(c.ws, v0 + j, lane, Va);
            f32x16 Wb = ksblock<D, KN>(c, j, xb, vb) - acc1;
            f32x16 Vb = zero16();
            mma_blk(Vb, Lop, Wb);
            stg(c.ws, v0 + NB + j, lane, Vb);
            // this wave loads these V blocks again in its next steps: a write-through store still on its way is not ordered
            // before a load of the same bytes, not even from the same wave (seen in the fp64 team kernel)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float zr = lds_f[c.L.z + 32 * j + rho(r, c.h)];
                vsa = fmaf(Va[r], Va[r], vsa);
                msa = fmaf(Va[r], zr, msa);
                vsb = fmaf(Vb[r], Vb[r], vsb);
                msb = fmaf(Vb[r], zr, msb);
            }
        }
        vsa = xhalf_sum(vsa); msa = xhalf_sum(msa);
        vsb = xhalf_sum(vsb); msb = xhalf_sum(msb);
        if (c.h == 0 && va) {
            const float var = c.sf2 - vsa;
            fm[qa] = msa; fv[qa] = var; yv[qa] = var + c.sn2;
        }
        if (c.h == 0 && vb) {
            const float var = c.sf2 - vsb;
            fm[qb] = msb; fv[qb] = var; yv[qb] = var + c.sn2;
        }
    }
    if (fcov) {
        // f*_cov = K_** - V^T V by 32 x 32 blocks (p <= q, mirrored), gpflow_models.py:245-263 (predict_f full_cov)
        coop_drain();                 // V blocks of other waves
        __syncthreads();
        int idx = 0;
        for (int p = 0; p < PC; ++p) {
            for (int q = p; q < PC; ++q, ++idx) {
                if ((idx & (NW - 1)) != c.w) continue;
                f32x16 Cb = zero16();
                f32x16 A = ldg(c.ws, c.cv0 + p * NB, lane), B = ldg(c.ws, c.cv0 + q * NB, lane);
                for (int k = 0; k + 1 < NB; ++k) {
                    const f32x16 nA = ldg(c.ws, c.cv0 + p * NB + k + 1, lane);
                    const f32x16 nB = ldg(c.ws, c.cv0 + q * NB + k + 1, lane);
                    mma_blk(Cb, A, B);
                    A = nA; B = nB;
                }
                mma_blk(Cb, A, B);
                const int qj = 32 * q + c.g;
                const bool vj = qj < c.P;
                float xq[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xq[d] = vj ? Xs[(size_t)qj * D + d] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pi = 32 * p + rho(r, c.h);
                    if (vj && pi < c.P) {
                        float r2 = 0.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            // difference of the raw coordinates, then the scale: (i,j) and (j,i) see exact negations
                            // whatever the compiler contracts, so the block is bitwise symmetric
                            const float df = (Xs[(size_t)pi * D + d] - xq[d]) * invl[d];
                            r2 = fmaf(df, df, r2);
                        }
                        float kf, gg;
                        kfun<KN>(r2, kf, gg);
                        const float v = c.sf2 * kf - Cb[r];
                        fcov[(size_t)pi * c.P + qj] = v;
                        if (p != q) fcov[(size_t)qj * c.P + pi] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cooperative tiles: the helper side (gpsat_coop.h).  A workgroup with no tile of its own attaches itself to a running
// tile (the one with the most work per attached workgroup, same XCD preferred: workgroup ids equal mod 8 share an L2) and
// takes part in the phases its owner opens -- the bulk queue of the sweep, the group queue of the gradient phase -- with
// all its waves.  It computes the same scaled coordinates from the same parameters as the owner, reads and writes the
// owner's workspace, and keeps nothing the owner waits for except its check-in (`active`).
// ---------------------------------------------------------------------------------------------
// (the context travels BY VALUE into these functions: a context whose address is taken would live in scratch memory in the
// kernel's hot loops too)
template <int D, int KN>
__device__ __noinline__ f32x2 evaluate_coop(Ctx<D, KN> c, bool want_grad) {
    evaluate<D, KN, true>(c, want_grad);
    f32x2 r = {c.sf2, c.sn2};
    return r;
}

struct HelpArgs {                // what a helper needs of the kernel arguments
    void* coop;
    float* ws;
    size_t ws_stride;
    const long long* obs_off;
    const float* X;
    int grid;
    int xcd_mode;                // developer: 0 any owner, 1 owners of this workgroup's XCD group only, 2 other groups only
};

// hp[0] owner attached to (-1: none)  hp[1] last phase sequence number taken part in  hp[2] tile staged in LDS (-1: none)
// hp[3] tile of the phase  hp[4] want_m  hp[5] decision of thread 0 for this episode (0: nothing, COOP_SWEEP, COOP_GRAD)
template <int D, int KN>
__device__ __noinline__ void helper_episode(Ctx<D, KN> c, const HelpArgs A) {
    Shared* sh = shared_state();
    gCoopCtl* ctls = as_gctl(A.coop);
    const int grid = A.grid;
    if (c.w == 0) {
        if (sh->hp[0] < 0) {
            // scan: lane l looks at the workgroups l, l + 64, ...; key = work per attached workgroup, same-XCD doubled
            int best = -1;
            for (int i = c.lane; i < grid; i += 64) {
                const int sc = __hip_atomic_load(&ctls[i].score, RLX_AGENT);
                if (sc <= 0 || i == (int)blockIdx.x) continue;
                if (A.xcd_mode == 1 && ((i ^ (int)blockIdx.x) & 7) != 0) continue;
                if (A.xcd_mode == 2 && ((i ^ (int)blockIdx.x) & 7) == 0) continue;
                const int hl = __hip_atomic_load(&ctls[i].helpers, RLX_AGENT);
                if (hl >= __hip_atomic_load(&ctls[i].hcap, RLX_AGENT)) continue;
                int key = (sc * (((i ^ (int)blockIdx.x) & 7) == 0 ? 2 : 1)) / (1 + hl);
                key = (min(key, 0xffff) << 12) | i;
                best = max(best, key);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) best = max(best, __shfl_xor(best, off));
            if (c.lane == 0 && best >= 0) {
                const int b = best & 0xfff;
                const int h = __hip_atomic_fetch_add(&ctls[b].helpers, 1, RLX_AGENT);
                if (h >= __hip_atomic_load(&ctls[b].hcap, RLX_AGENT)) {
                    __hip_atomic_fetch_add(&ctls[b].helpers, -1, RLX_AGENT);
                } else {
                    sh->hp[0] = b;
                    sh->hp[1] = -1;           // any open phase may be joined
                }
            }
        }
        if (c.lane == 0) {
            int decision = 0;
            const int b = sh->hp[0];
            if (b >= 0) {
                gCoopCtl* ctl = ctls + b;
                const unsigned long long pav = __hip_atomic_load(&ctl->pa, RLX_AGENT);
                const unsigned wd = PA_PHASE(pav);
                const int kind = (int)(wd & 3u), seq = (int)(wd >> 2);
                if (kind == COOP_RELEASED || __hip_atomic_load(&ctl->score, RLX_AGENT) <= 0) {
                    __hip_atomic_fetch_add(&ctl->helpers, -1, RLX_AGENT);
                    sh->hp[0] = -1;
                } else if ((kind == COOP_SWEEP || kind == COOP_GRAD) && seq != sh->hp[1]) {
                    // check in: count + 1 while the whole word is what was read (a closed or reopened phase, or another helper's
                    // check-in, makes it fail: look again in the next episode)
                    unsigned long long expect = pav;
                    if (__hip_atomic_compare_exchange_strong(&ctl->pa, &expect, pav + 1ull, RLX_AGENT_CAS)) {
                        decision = kind;
                        sh->hp[1] = seq;
                        sh->hp[3] = __hip_atomic_load(&ctl->tile, RLX_AGENT);
                        sh->hp[4] = __hip_atomic_load(&ctl->want_m, RLX_AGENT);
                        for (int i = 0; i < D + 2; ++i) sh->theta[i] = gld_d(&ctl->theta[i]);
                    }
                }
            }
            if (!decision) __builtin_amdgcn_s_sleep(64);
            sh->hp[5] = decision;
        }
    }
    __syncthreads();
    const int kind = sh->hp[5];
    if (kind == 0) return;
    const int b = sh->hp[0], t = sh->hp[3];
    const bool want_m = sh->hp[4] != 0;
    // this workgroup's context becomes the owner's tile
    c.helper = true;
    c.ctl = ctls + b;
    c.ws = A.ws + (size_t)b * A.ws_stride;
    c.zg = as_gfloat(c.ws + (size_t)(c.zb - 8) * BLK);
    c.ag = c.zg + 4 * BLK;
    const long long o0 = A.obs_off[t], o1 = A.obs_off[t + 1];
    c.N = (int)(o1 - o0);
    c.P = 0;
    c.NB = (c.N + 31) / 32;
    c.Npad = c.NB * 32;
    c.dT0 = c.NB * c.NB;
    c.vs0 = c.dT0 + c.NB;
    c.cv0 = c.vs0 + NW * 2 * c.NB;
    c.gp0 = c.vs0 * (BLK * 4);
    const bool restage = sh->hp[2] != t;          // every thread reads the marker ...
    __syncthreads();                              // ... before thread 0 moves it
    if (restage) {
        for (int idx = c.tid; idx < c.Npad; idx += NT) {
            const bool v = idx < c.N;
#pragma unroll
            for (int d = 0; d < D; ++d) lds_f[c.L.xs + d * c.Npad + idx] = v ? A.X[(size_t)(o0 + idx) * D + d] : 0.f;
        }
        if (c.tid == 0) sh->hp[2] = t;
    }
    __syncthreads();
    {   // the owner's scaled coordinates: same operations on the same values (evaluate)
        float invl[D];
#pragma unroll
        for (int d = 0; d < D; ++d) invl[d] = (float)((double)KScale<KN>::c / sh->theta[d]);
        c.sf2 = (float)sh->theta[D];
        c.sn2 = (float)sh->theta[D + 1];
        for (int idx = c.tid; idx < c.Npad; idx += NT) {
#pragma unroll
            for (int d = 0; d < D; ++d) lds_f[c.L.xsc + d * c.Npad + idx] = lds_f[c.L.xs + d * c.Npad + idx] * invl[d];
        }
        if (kind == COOP_GRAD)
            for (int idx = c.tid; idx < c.Npad; idx += NT) lds_f[c.L.alpha + idx] = gld_f(c.ag + idx);
    }
    __syncthreads();
    if (c.tid == 0) COOP_STAT(ctls + blockIdx.x, 1);
    if (kind == COOP_SWEEP) {
        if (!pt_bulk_loop<D, KN, true>(c, want_m, true) && c.lane == 0) {
            __hip_atomic_store(&c.ctl->fail, 2, RLX_AGENT);
            COOP_STAT(ctls + blockIdx.x, 6);
        }
    } else {
        grad_loop<D, KN, true>(c);
    }
    coop_drain();
    __syncthreads();
    if (c.tid == 0) __hip_atomic_fetch_add(&c.ctl->pa, ~0ull, RLX_AGENT);          // check out: count - 1
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// the persistent kernel
// ---------------------------------------------------------------------------------------------
template <int D, int KN>
__global__ void __launch_bounds__(NT, GPSAT_MIN_WG) gp_tile_kernel(const KernelArgs A) {
    constexpr int H = D + 2;
    Ctx<D, KN> c;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.w = c.tid >> 6;
    c.h = c.lane >> 5;
    c.g = c.lane & 31;
    const int NPmax = A.NBmax * 32;
    Shared* sh = shared_state();
    int off = SHARED_FLOATS;
    c.L.xs = off; off += D * NPmax;
    c.L.xsc = off; off += D * NPmax;
    c.L.y = off; off += NPmax;
    c.L.z = off; off += NPmax;
    c.L.alpha = off; off += NPmax;
    c.L.LT = off; off += 4 * BLK;
    c.L.U01 = off; off += 2 * BLK;
    c.L.Wh = off; off += 4 * BLK;          // column wave: finished k-loop of the next group 0, parked for the chain wave
    c.L.Ad = off; off += 32 * 33 + 3;       // 1059 -> keep the next offsets 16-B aligned
    off = (off + 3) & ~3;
    c.L.tmp = off; off += 32;
    c.L.piv = off; off += 64;
    float* const ws_own = A.ws + (size_t)blockIdx.x * A.ws_stride;
    c.ws = ws_own;
    c.zb = (int)(A.ws_stride / BLK) - 1;            // last block of the workgroup's workspace: zeros
    if (c.w == 0) { stg(c.ws, c.zb, c.lane, zero16()); coop_drain(); }     // (read behind the barrier at the top of the tile loop)
    // cooperative tiles: this workgroup's control block (as an owner); z and alpha of a cooperative evaluation live in
    // the 8 blocks in front of the zero block
    // Cooperative tiles are compiled into the 8-wave build only: the 4-wave build runs two workgroups per CU, where a helper
    // gives nothing, and without the two out-of-line call sites its tile loop keeps half as many
    // registers in scratch.  Launches with fewer tiles than CUs take the 8-wave build (gpsat_capi.cpp).
    constexpr bool COOP_BUILD = (NW == 8);
    const bool coop_on = COOP_BUILD && A.coop != nullptr;
    gCoopCtl* const ctl_own = as_gctl(A.coop) + blockIdx.x;       // dereferenced only when coop_on
    if (c.tid == 0) { sh->hp[0] = -1; sh->hp[1] = -1; sh->hp[2] = -1; sh->coop_now = 0; sh->coop_seq = 0; }
    c.prof = sh->prof;
    c.trace = nullptr;
#ifdef GPSAT_PROFILE
    if (blockIdx.x == 0 && A.prof) c.trace = A.prof + 64;
    if (c.tid < NW) sh->tcnt[c.tid] = 0;
    int prof_ntiles = 0;                          // the evaluations of the THIRD tile of workgroup 0 are traced
    if (c.tid < NW * 16) sh->prof[c.tid] = 0ull;
    const unsigned long long prof_k0 = __builtin_amdgcn_s_memtime(), prof_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    OptCfg o;
    o.optimiser = A.optimiser; o.max_iter = A.max_iter; o.max_ls = A.max_ls; o.want_grad_out = A.grad != nullptr;
    o.ftol = A.ftol; o.gtol = A.gtol; o.adam_lr = A.adam_lr; o.noise_rel = A.noise_rel;

    const bool sliced = A.seg_cost > 0;
    for (;;) {
        __syncthreads();
        HelpArgs ha;
        ha.coop = A.coop; ha.ws = A.ws; ha.ws_stride = A.ws_stride; ha.obs_off = A.obs_off; ha.X = A.X;
        ha.grid = (int)gridDim.x;
        ha.xcd_mode = (A.coop_force >> 2) & 3;
        if (sliced && coop_on) {
            // claim a ring slot; while it is empty (the tail of the batch: fewer unfinished tiles than workgroups) help a
            // running tile instead of spinning
            if (c.tid == 0) sh->hp[7] = (int)ring_claim(A);
            for (int spins = 0;; ++spins) {
                if (c.tid == 0) sh->tile = (spins > (1 << 22)) ? -1 : ring_look(A, (unsigned)sh->hp[7]);
                __syncthreads();
                if (sh->tile != -2) break;
                helper_episode<D, KN>(c, ha);
            }
        } else if (c.tid == 0) {
            if (sliced) {
#ifdef GPSAT_PROFILE
                const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
                sh->tile = ring_pop(A);
#ifdef GPSAT_PROFILE
                sh->prof[13] += __builtin_amdgcn_s_memtime() - tw0;      // time this workgroup waited for a tile (scripts/tail_profile.py)
#endif
            } else {
                const int slot = atomicAdd(A.queue, 1);
                sh->tile = slot < A.T ? A.order[slot] : -1;
            }
        }
        __syncthreads();
        int entry = sh->tile;
        if (entry == -1 && coop_on && !sliced) {
            // no tile left for this workgroup: help the tiles that are still running until the last one has finished
            for (;;) {
                if (c.tid == 0) sh->hp[6] = __hip_atomic_load(A.coop_live, RLX_AGENT);
                __syncthreads();
                if (sh->hp[6] <= 0) break;
                helper_episode<D, KN>(c, ha);
            }
        }
        if (entry == -1) break;
#ifdef GPSAT_PROFILE
        if (c.tid == 0) sh->tron = (prof_ntiles == 2);
        ++prof_ntiles;
#endif
        const int t = entry & 0x7fffffff;
        const bool resumed = entry < 0;           // bit 31: the tile's optimiser state is in A.state
        const long long o0 = A.obs_off[t], o1 = A.obs_off[t + 1];
        const long long p0 = A.pred_off[t], p1 = A.pred_off[t + 1];
        c.N = (int)(o1 - o0);
        c.P = (int)(p1 - p0);
        c.NB = (c.N + 31) / 32;
        c.Npad = c.NB * 32;
        const int NB = c.NB;
        c.dT0 = NB * NB;
        c.vs0 = c.dT0 + NB;
        c.cv0 = c.vs0 + NW * 2 * NB;
        c.gp0 = c.vs0 * (BLK * 4);
        if (c.N == 0) {
            if (c.tid == 0) {
                A.status[t] = 4; A.n_eval[t] = 0; A.nll[t] = 0.0;
                if (A.n_iter) A.n_iter[t] = 0;
                for (int i = 0; i < H; ++i) {
                    A.theta[(size_t)t * H + i] = A.theta0[(size_t)t * H + i];
                    if (A.grad) A.grad[(size_t)t * H + i] = 0.0;
                }
            }
            // prior prediction for an empty tile
            for (long long q = p0 + c.tid; q < p1; q += NT) {
                const float sf2 = (float)A.theta0[(size_t)t * H + D], sn2 = (float)A.theta0[(size_t)t * H + D + 1];
                A.f_mean[q] = 0.f; A.f_var[q] = sf2; A.y_var[q] = sf2 + sn2;
            }
            if (A.f_cov) {
                // prior covariance K_** of an empty tile
                const int Pn = (int)(p1 - p0);
                const float sf2 = (float)A.theta0[(size_t)t * H + D];
                for (long long e = c.tid; e < (long long)Pn * Pn; e += NT) {
                    const int i = (int)(e / Pn), j = (int)(e % Pn);
                    float r2 = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float il = (float)((double)KScale<KN>::c / A.theta0[(size_t)t * H + d]);
                        const float df = (A.Xs[(size_t)(p0 + i) * D + d] - A.Xs[(size_t)(p0 + j) * D + d]) * il;
                        r2 = fmaf(df, df, r2);
                    }
                    float kf, gg;
                    kfun<KN>(r2, kf, gg);
                    A.f_cov[A.cov_off[t] + e] = sf2 * kf;
                }
            }
            if (sliced && c.tid == 0) __hip_atomic_fetch_add(&A.ring_ctl[32], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (coop_on && c.tid == 0) __hip_atomic_fetch_add(A.coop_live, -1, RLX_AGENT);
            continue;
        }
        // this workgroup owns the tile: its own workspace and control block (it may have helped another tile while it
        // waited for this one: it lets go of it)
        if (coop_on && c.tid == 0 && sh->hp[0] >= 0) {
            __hip_atomic_fetch_add(&(as_gctl(A.coop) + sh->hp[0])->helpers, -1, RLX_AGENT);
            sh->hp[0] = -1;
        }
        c.helper = false;
        c.ws = ws_own;
        c.ctl = ctl_own;
        c.zg = as_gfloat(ws_own + (size_t)(c.zb - 8) * BLK);
        c.ag = c.zg + 4 * BLK;
        // ---- stage tile data into LDS (SoA coordinates), zero padding
        for (int idx = c.tid; idx < c.Npad; idx += NT) {
            const bool v = idx < c.N;
#pragma unroll
            for (int d = 0; d < D; ++d) lds_f[c.L.xs + d * c.Npad + idx] = v ? A.X[(size_t)(o0 + idx) * D + d] : 0.f;
            lds_f[c.L.y + idx] = v ? A.y[o0 + idx] : 0.f;
            lds_f[c.L.z + idx] = 0.f;
            lds_f[c.L.alpha + idx] = 0.f;
        }
        if (resumed) {
            // The state was written by another workgroup, possibly on another XCD (whose L2 is not coherent with this one).
            // It travels through agent-scope atomic word accesses, which go to memory past the caches: no cache-wide
            // write-back / invalidate (an agent-scope fence costs every workgroup of the XCD its L2 contents).  Every wave of
            // the writer had drained its stores (s_waitcnt vmcnt(0)) and met the workgroup barrier before one lane published
            // the ring entry that this workgroup's thread 0 has polled (sc1 load) ahead of the barrier above; every load of
            // the state is an sc1 load to registers (MI355X_MICROARCH.md, inter-workgroup visibility: valid forms).
            const unsigned* src = A.state + (size_t)t * A.state_words;
            unsigned* dst = reinterpret_cast<unsigned*>(sh);
            for (int i = c.tid; i < A.state_words; i += NT)
                dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (c.tid == 0) {
            sh->n_eval = 0; sh->n_eval_opt = 0; sh->status = 5; sh->iter = 0; sh->hist_n = 0; sh->hist_pos = 0;
            sh->last_dec = 1e300;
            sh->fail = 0;
            for (int i = 0; i < H; ++i) {
                const double lo = A.lo[(size_t)t * H + i], hi = A.hi[(size_t)t * H + i];
                const bool box = (lo == lo) && (hi == hi) && (fabs(lo) < 1e300) && (fabs(hi) < 1e300);
                sh->box[i] = box ? 1 : 0;
                sh->lo[i] = lo; sh->hi[i] = hi;
                sh->shift[i] = (!box && i == D + 1) ? 1e-6 : 0.0;   // GPflow likelihood-variance lower bound
                sh->trainable[i] = A.trainable[i] ? 1 : 0;
                sh->theta[i] = A.theta0[(size_t)t * H + i];
                sh->u[i] = u_of_theta(sh, i, sh->theta[i]);
                sh->m1[i] = 0.0; sh->m2[i] = 0.0;
            }
            const bool optim = (o.optimiser != 0 && o.max_iter > 0);
            sh->phase = optim ? PH_INIT : PH_FINAL;
            sh->want_grad = optim ? 1 : o.want_grad_out;
        }
        const bool helpable = coop_on && NB >= A.coop_min_nb;
        if (c.tid == 0) {
            sh->hp[2] = -1;                        // the coordinates in LDS are this tile's, not a helped one's
            if (coop_on) sh->coop_seq = (int)(PA_PHASE(__hip_atomic_load(&ctl_own->pa, RLX_AGENT)) >> 2);
            if (helpable) {
                // helpers wanted: one per coop_hdiv block columns (the bulk queue of a panel has NB / 2 groups), 7 at most
                __hip_atomic_store(&ctl_own->tile, t, RLX_AGENT);
                __hip_atomic_store(&ctl_own->hcap, min(7, max(1, NB / A.coop_hdiv)), RLX_AGENT);
                __hip_atomic_store(&ctl_own->pa, PA_MAKE(sh->coop_seq, COOP_CLOSED), RLX_AGENT);
                __hip_atomic_store(&ctl_own->score, NB, RLX_AGENT);
            }
        }
        __syncthreads();

        // ================= evaluate / advance loop (one inlined evaluate call site per mode) =================
        const int seg_evals = sliced ? max(1, A.seg_cost / (NB * NB * NB)) : 0x7fffffff;
        bool suspended = false;
        for (int nseg = 1;; ++nseg) {
            if (helpable) {
                // cooperative evaluation when helpers are attached (they may still leave: nothing waits for them)
                if (c.tid == 0) sh->coop_now = ((A.coop_force & 1) || __hip_atomic_load(&ctl_own->helpers, RLX_AGENT) > 0) ? 1 : 0;
                __syncthreads();
            }
            if (helpable && sh->coop_now) {
                const f32x2 sv = evaluate_coop<D, KN>(c, sh->want_grad != 0);
                c.sf2 = sv[0]; c.sn2 = sv[1];
            } else {
                evaluate<D, KN, false>(c, sh->want_grad != 0);
            }
            if (c.tid == 0) opt_advance(sh, H, o);
            __syncthreads();
            if (sh->phase == PH_EXIT) break;
            // time slice used up while the optimiser goes on (the final evaluation + prediction are never split off:
            // prediction needs this workgroup's factorisation)
            if (nseg >= seg_evals && sh->phase != PH_FINAL) {
                // ... unless no other tile is waiting in the ring: the slice would only hand this tile to a workgroup that
                // waits for work (state through device memory for nothing) and shake off the tile's helpers
                if (c.tid == 0) sh->hp[6] = ring_waiting_tiles(A);
                __syncthreads();
                if (sh->hp[6] > 0) { suspended = true; break; }
            }
        }
        if (helpable && c.tid == 0) {
            // no more cooperative phases from this tile: its helpers look elsewhere (the prediction is the owner's alone)
            __hip_atomic_store(&ctl_own->score, 0, RLX_AGENT);
            __hip_atomic_store(&ctl_own->pa, PA_MAKE(sh->coop_seq, COOP_RELEASED), RLX_AGENT);
        }
        if (suspended) {
            unsigned* dst = A.state + (size_t)t * A.state_words;
            const unsigned* src = reinterpret_cast<const unsigned*>(sh);
            for (int i = c.tid; i < A.state_words; i += NT)
                __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // EVERY storing wave drains its own stores (a workgroup-scope fence emits no vmcnt wait on gfx950; inline asm
            // so that no compiler pass can drop or move it), THEN the barrier, THEN one lane publishes the ring entry
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c.tid == 0) ring_push(A, t);
            continue;
        }

#ifdef GPSAT_DUMP
        if (A.dump) {              // diagnostic build: what the last evaluation left behind, per tile
            __syncthreads();
            float* dp = A.dump + (size_t)t * A.dump_stride;
            const int nblk = NB * NB + NB;
            for (int bq = c.w; bq < nblk; bq += NW) { const f32x16 v = ldg(c.ws, bq, c.lane); stg(dp, bq, c.lane, v); }
            for (int i = c.tid; i < c.Npad; i += NT) {
                dp[(size_t)nblk * BLK + i] = lds_f[c.L.z + i];
                dp[(size_t)nblk * BLK + c.Npad + i] = lds_f[c.L.alpha + i];
            }
            if (c.tid == 0) *reinterpret_cast<double*>(dp + (size_t)nblk * BLK + 2 * c.Npad) = sh->logdet;
            // the chain's staged partial sums (DBG_TP): the 8 blocks in front of the zero block
            for (int i = c.tid; i < 8 * BLK; i += NT)
                dp[(size_t)nblk * BLK + 2 * c.Npad + 16 + i] = c.ws[(size_t)(c.zb - 8) * BLK + i];
            coop_drain();
            __syncthreads();
        }
#endif
        // ================= outputs + prediction from the factorisation at the accepted parameters
        if (c.tid == 0) {
            int st = sh->status;
            if (sh->fail) st = (sh->nll == sh->nll) ? 2 : 3;
            A.status[t] = st;
            A.n_eval[t] = sh->n_eval_opt;
            if (A.n_iter) A.n_iter[t] = sh->iter;
            A.nll[t] = sh->fail ? __builtin_nan("") : sh->nll;
            for (int i = 0; i < H; ++i) {
                A.theta[(size_t)t * H + i] = sh->theta[i];
                if (A.grad) A.grad[(size_t)t * H + i] = sh->fail ? __builtin_nan("") : sh->gth[i];
            }
        }
        if (c.P > 0) {
            if (!sh->fail) {
                float invl[D];
#pragma unroll
                for (int d = 0; d < D; ++d) invl[d] = (float)((double)KScale<KN>::c / sh->theta[d]);
                predict_tile<D, KN>(c, A.Xs + (size_t)p0 * D, A.f_mean + p0, A.f_var + p0, A.y_var + p0, invl,
                                    A.f_cov ? A.f_cov + A.cov_off[t] : nullptr);
            } else {
                for (long long q = p0 + c.tid; q < p1; q += NT) {
                    A.f_mean[q] = __builtin_nanf(""); A.f_var[q] = __builtin_nanf(""); A.y_var[q] = __builtin_nanf("");
                }
                if (A.f_cov)
                    for (long long q = A.cov_off[t] + c.tid; q < A.cov_off[t + 1]; q += NT) A.f_cov[q] = __builtin_nanf("");
            }
        }
        if (sliced && c.tid == 0) __hip_atomic_fetch_add(&A.ring_ctl[32], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (coop_on && c.tid == 0) __hip_atomic_fetch_add(A.coop_live, -1, RLX_AGENT);
    }
#ifdef GPSAT_PROFILE
    __syncthreads();
    if (A.prof && c.tid == 0 && blockIdx.x < 1024) {       // when did this workgroup start and run out of work (100 MHz ticks)
        A.prof[64 + 8 * 1024 + blockIdx.x] = prof_r0;
        A.prof[64 + 8 * 1024 + 1024 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    if (blockIdx.x == 0 && c.tid == 0) {       // whole-kernel span of workgroup 0 in both clocks: s_memtime ticks per 100 MHz tick
        sh->prof[14] = __builtin_amdgcn_s_memtime() - prof_k0;
        sh->prof[15] = __builtin_amdgcn_s_memrealtime() - prof_r0;
    }
    __syncthreads();
    if (A.prof && c.tid < NW * 16) atomicAdd(&A.prof[c.tid], sh->prof[c.tid]);
#endif
}

size_t shared_bytes(int D, int NBmax) {
    const size_t NP = (size_t)NBmax * 32;
    size_t fl = (size_t)SHARED_FLOATS + 2 * D * NP + 3 * NP + 10 * BLK + 32 * 33 + 3 + 4 + 32 + 64;
    return (fl * sizeof(float) + 15) & ~size_t(15);
}

size_t workspace_floats_per_wg(int NBmax, int PCcov) {
    // U/M square + DinvT + per-wave V scratch (2 chunks) [+ V of all chunks for the full covariance, one spare chunk
    // for the odd partner] + one block of zeros
    const size_t cov = PCcov > 0 ? (size_t)(PCcov + 1) * NBmax : 0;
    // ... 8 blocks for z and alpha of a cooperative evaluation (gpsat_coop.h), and the zero block
    return (size_t)BLK * ((size_t)NBmax * NBmax + (size_t)NBmax + (size_t)NW * 2 * NBmax + cov + 8 + 1);
}

template <int D, int KN>
static hipError_t launch_one(const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gp_tile_kernel<D, KN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gp_tile_kernel<D, KN>), dim3(grid), dim3(NT), smem, stream, a);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_d(const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
    switch (a.kernel) {
        case 0: return launch_one<D, 0>(a, grid, smem, stream);
        case 1: return launch_one<D, 1>(a, grid, smem, stream);
        case 2: return launch_one<D, 2>(a, grid, smem, stream);
        case 3: return launch_one<D, 3>(a, grid, smem, stream);
        default: return hipErrorInvalidValue;
    }
}

// one specialised kernel per (input dimension, covariance function): the covariance is inlined into the
// K-block / contraction code, so specialising keeps each kernel's code (and its I-cache footprint) small
hipError_t launch_tiles(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
    switch (D) {
        case 1: return launch_d<1>(a, grid, smem, stream);
        case 2: return launch_d<2>(a, grid, smem, stream);
        case 3: return launch_d<3>(a, grid, smem, stream);
        case 4: return launch_d<4>(a, grid, smem, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace GPSAT_VNS

size_t GPSAT_VFN(shared_bytes)(int D, int NBmax) { return GPSAT_VNS::shared_bytes(D, NBmax); }
int GPSAT_VFN(state_words)() { return GPSAT_VNS::SHARED_FLOATS; }
size_t GPSAT_VFN(workspace_floats_per_wg)(int NBmax, int PCcov) { return GPSAT_VNS::workspace_floats_per_wg(NBmax, PCcov); }
hipError_t GPSAT_VFN(launch_tiles)(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
    return GPSAT_VNS::launch_tiles(D, a, grid, smem, stream);
}

}  // namespace gpsat
