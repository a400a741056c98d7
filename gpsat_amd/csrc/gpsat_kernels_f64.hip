// gpsat_kernels_f64.hip -- fp64 variant of the persistent local-expert tile kernel (gfx950).
//
// Same algorithm, same data-layout idea and the same on-device optimiser as the fp32 kernel
// (gpsat_kernels.hip), on 16x16 fp64 blocks and v_mfma_f64_16x16x4_f64:
//   * "acc layout" of a block = the f64 MFMA accumulator: lane l = 16*q + g owns column g and the rows
//     q + 4r, r = 0..3 (NOT the f32 row map); 4 doubles per lane, 32 contiguous bytes per lane in memory;
//   * a stored block S acts as S^T as the A operand and as S as the B operand (MFMA step s contracts the rows
//     q + 4s), an accumulator is directly a B operand, so every product is S_A^T * S_B;
//   * K = U^T U (upper), M = L^-1 (lower), (K^-1)_ab = sum_c M_ca^T M_cb contracted on the fly with dK/dtheta.
// This first fp64 version keeps the simple schedule (one block row per step, one workgroup barrier pair per
// row, one block column of M per wave); it exists for the reference's native precision (GPflow default_float,
// fp64 throughout) -- 1e-6-level known-answer parity and BASELINE config 5 (N = 2000,
// predict-only with loaded hyper-parameters).  Maths: SURVEY.md Appendix A, GPSat/models/pure_python_gpr.py:485-498.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gpsat_kernels.h"
#include "gpsat_ring.h"

namespace gpsat {
// Two builds are linked (as for the fp32 kernels): the default, 8 waves per workgroup and one workgroup per CU, and
// -DGPSAT_F64_W4, 4 waves and two workgroups per CU for tiles whose LDS fits twice (a second TILE per CU overlaps the
// serial diagonal work of the first).
#ifdef GPSAT_F64_W4
#define GPSAT_F64_NW 4
#define F64NS f64k4
#define F64FN(name) name##_w4
#define F64_MIN_WG 2
#else
#define F64NS f64k
#define F64FN(name) name
#define F64_MIN_WG 1
#endif
namespace F64NS {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

extern __shared__ __attribute__((aligned(16))) double lds_d[];

#ifndef GPSAT_F64_NW
#define GPSAT_F64_NW 8
#endif
#define GPSAT_NW GPSAT_F64_NW     // one workgroup per CU (LDS): 8 waves = 2 per SIMD hide the operand latency
#include "gpsat_opt.h"
#undef GPSAT_NW

constexpr int BS = 16;         // block size
constexpr int BLK = 256;       // doubles per block

// Diagnostic build only (-DGPSAT_PROFILE, scripts/phase_profile_f64.py): per-wave cycle counters per code segment in LDS,
// flushed to KernelArgs::prof.  No stamp executes in the product build.
#ifdef GPSAT_PROFILE
#define PROF_BEGIN() unsigned long long prof_t_ = __builtin_amdgcn_s_memtime()
#define PROF_END(c_, slot_)                                                                              \
    do {                                                                                                 \
        unsigned long long t1_ = __builtin_amdgcn_s_memtime();                                           \
        if ((c_).lane == 0) reinterpret_cast<Shared*>(lds_d)->prof[(c_).w * 16 + (slot_)] += t1_ - prof_t_; \
        prof_t_ = t1_;                                                                                   \
    } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_END(c_, slot_) do {} while (0)
#endif

__device__ __forceinline__ int rowof(int r, int q) { return q + 4 * r; }

// Workspace blocks.  A tile run by ONE workgroup moves them with plain global loads / stores (the CU's own L1 / L2 serve
// them).  A tile run by a TEAM of workgroups (below) moves them device-coherently: buffer_load / buffer_store ... sc1
// (write-through, past L1), every hand-over behind a drain -- MI355X_MICROARCH.md, "inter-workgroup visibility".  A tile is
// run in one mode from its first to its last evaluation, so the two kinds of access never meet on the same bytes.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool TEAM>
__device__ __forceinline__ f64x4 ldg_t(const double* __restrict__ ws, int blk, int lane) {
    if (TEAM) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ws), 0, 0x7fffffff, 0x00020000);
        const int so = blk * (BLK * 8), vo = lane * 32;
        const f64x2 a = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 16));
        const f64x2 b = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, vo + 16, so, 16));
        f64x4 v = {a[0], a[1], b[0], b[1]};
        return v;
    }
    const f64x2* p = reinterpret_cast<const f64x2*>(ws + (size_t)blk * BLK + lane * 4);
    const f64x2 a = p[0], b = p[1];
    f64x4 v = {a[0], a[1], b[0], b[1]};
    return v;
}

template <bool TEAM>
__device__ __forceinline__ void stg_t(double* __restrict__ ws, int blk, int lane, const f64x4& v) {
    f64x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    if (TEAM) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
        const int so = blk * (BLK * 8), vo = lane * 32;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), r, vo, so, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), r, vo + 16, so, 16);
        return;
    }
    f64x2* p = reinterpret_cast<f64x2*>(ws + (size_t)blk * BLK + lane * 4);
    p[0] = a;
    p[1] = b;
}
#define ldg ldg_t<TEAM>
#define stg stg_t<TEAM>

__device__ __forceinline__ f64x4 ldl(int off, int lane) {
    const f64x2* p = reinterpret_cast<const f64x2*>(lds_d + off + lane * 4);
    const f64x2 a = p[0], b = p[1];
    f64x4 v = {a[0], a[1], b[0], b[1]};
    return v;
}

__device__ __forceinline__ void stl(int off, int lane, const f64x4& v) {
    f64x2* p = reinterpret_cast<f64x2*>(lds_d + off + lane * 4);
    f64x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    p[0] = a;
    p[1] = b;
}

// acc += S_A^T * S_B   (4 x v_mfma_f64_16x16x4_f64)
__device__ __forceinline__ void mma_blk(f64x4& acc, const f64x4& a, const f64x4& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
}

__device__ __forceinline__ f64x4 zero4() { f64x4 z = {0.0, 0.0, 0.0, 0.0}; return z; }

__device__ __forceinline__ double qsum(double v) {      // sum over the 4 row groups of a column (lanes g, g+16, g+32, g+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <int KERN>
__device__ __forceinline__ void kfun(double r2, double& kf, double& gg) {
    if (KERN == 0) {
        kf = exp(-0.5 * r2);
        gg = kf;
    } else {
        const double r = sqrt(fmax(r2, 1e-36));
        if (KERN == 1) {
            kf = exp(-r);
            gg = kf / r;
        } else if (KERN == 2) {
            const double s = 1.7320508075688772 * r, e = exp(-s);
            kf = (1.0 + s) * e;
            gg = 3.0 * e;
        } else {
            const double s = 2.23606797749979 * r, e = exp(-s);
            kf = (1.0 + s + s * s * (1.0 / 3.0)) * e;
            gg = (5.0 / 3.0) * (1.0 + s) * e;
        }
    }
}

struct Lay { int xsc, y, z, alpha, Ad, LT, tmp, Pn, tp4, PnLA, tpLA; };   // double offsets into lds_d

// ---------------------------------------------------------------------------------------------
// Teams (large fp64 tiles): G workgroups run ONE tile together, bulk-synchronously -- the kernel's phases as they are, every
// workgroup barrier between them replaced by a team barrier in device memory, the loops over block columns dealt over the
// team's G * NW waves, and what the waves of one workgroup exchange through LDS (the panel's diagonal region, z, alpha)
// exchanged through the workspace of member 0 instead.  Member 0 (the owner) runs the optimiser, the serial part of every
// panel and the prediction; the others follow its commands.  The team is fixed by the host for the whole launch
// (KernelArgs::team_size), so a tile's workspace is only ever touched device-coherently (ldg_t<true> / stg_t<true>).
// Per item the arithmetic is what one workgroup does, per-column updates of alpha keep their order (one writer per column
// and panel, panels separated by team barriers), the gradient's partial sums are added in a fixed order: a team returns the
// bits one workgroup returns.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) double gdouble;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

enum { TEAM_EVAL = 1, TEAM_TILE = 2, TEAM_EXIT = 3 };

struct TeamCtl {                 // 256 B per team, zeroed by the host before the launch; agent-scope atomics only
    int bar;                     // team barrier: arrivals so far (monotone)
    int fail;                    // the running evaluation has failed (owner, before the barrier that follows the serial part)
    int cmd;                     // owner -> members, read behind the command barrier
    int tile;
    int want_grad;
    int timeout;                 // a barrier gave up (never by design)
    int pad0[2];
    double theta[8];
    int pad1[40];
};
static_assert(sizeof(TeamCtl) == 256, "TeamCtl is 256 bytes");
typedef __attribute__((address_space(1))) TeamCtl gTeamCtl;

__device__ __forceinline__ double gld_d(const gdouble* p) { return __longlong_as_double((long long)__hip_atomic_load((const gu64*)p, RLX_AGENT)); }
__device__ __forceinline__ void gst_d(gdouble* p, double v) { __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), RLX_AGENT); }

template <int D, int KN>
struct Ctx {
    Lay L;
    double* ws;
    int zb, dT0, vs0, cv0;
    int N, NB, Npad, P;
    int tid, lane, w, q, g;
    double sf2, sn2;
    // team: this wave's index among the team's G * NW waves, their number, the team's control block, and the exchange areas
    // in the owner's workspace (panel diagonal region 10 blocks + 64 doubles, z, alpha, the gradient's partial sums)
    int vw, nwt, member, G;
    gTeamCtl* tc;
    int pn0;                     // block index of the exchanged diagonal region
    gdouble *tpg, *zg, *ag;
    int gp0;                     // byte offset of the gradient phase's per-item partial sums (aliases the prediction scratch)
};

// Every wave has drained its stores, every workgroup of the team has arrived: data stored sc1 before the barrier is read
// (sc1) behind it by any member.  One workgroup (G == 1): a plain workgroup barrier.  A barrier that gives up (never by
// design; bounded so that a lost workgroup cannot hang the GPU) marks the evaluation failed.
// `long_wait`: the command barrier, at which the members wait while the owner predicts (seconds for many prediction points).
template <bool TEAM, int D, int KN>
__device__ __forceinline__ void team_barrier(Ctx<D, KN>& c, bool long_wait = false) {
    if (!TEAM) { __syncthreads(); return; }
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (c.tid == 0) {
        sh->hp[0] += 1;                                       // barriers passed by this workgroup
        const int target = sh->hp[0] * c.G;
        __hip_atomic_fetch_add(&c.tc->bar, 1, RLX_AGENT);
        for (int spins = 0; __hip_atomic_load(&c.tc->bar, RLX_AGENT) < target; ++spins) {
            __builtin_amdgcn_s_sleep(4);
            if (long_wait && spins > (1 << 16)) __builtin_amdgcn_s_sleep(127);      // ~1 us per poll: minutes before it gives up
            if (spins > (long_wait ? (1 << 27) : (1 << 23)) || ((spins & 255) == 255 && __hip_atomic_load(&c.tc->timeout, RLX_AGENT))) {
                __hip_atomic_store(&c.tc->timeout, 1, RLX_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

template <int D, int KN>
__device__ __forceinline__ f64x4 kblock(const Ctx<D, KN>& c, int bi, int bj) {
    const int qc = BS * bj + c.g;
    double xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = lds_d[c.L.xsc + d * c.Npad + qc];
    f64x4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = BS * bi + rowof(r, c.q);
        double r2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) { const double df = lds_d[c.L.xsc + d * c.Npad + p] - xq[d]; r2 = fma(df, df, r2); }
        double kf, gg;
        kfun<KN>(r2, kf, gg);
        double v = ((p < c.N) && (qc < c.N)) ? c.sf2 * kf : 0.0;
        if (p == qc) v = (p < c.N) ? (v + c.sn2) : 1.0;
        out[r] = v;
    }
    return out;
}

template <int D, int KN>
__device__ __forceinline__ f64x4 ksblock(const Ctx<D, KN>& c, int bj, const double (&xq)[D], bool qvalid) {
    f64x4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = BS * bj + rowof(r, c.q);
        double r2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) { const double df = lds_d[c.L.xsc + d * c.Npad + p] - xq[d]; r2 = fma(df, df, r2); }
        double kf, gg;
        kfun<KN>(r2, kf, gg);
        out[r] = ((p < c.N) && qvalid) ? c.sf2 * kf : 0.0;
    }
    return out;
}

template <int D, int KN>
__device__ __forceinline__ void contract(const Ctx<D, KN>& c, const f64x4& kinv, int ba, int bb, double wgt,
                                         double (&accl)[D], double& accsf, double& accsn) {
    const int qc = BS * bb + c.g;
    double xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = lds_d[c.L.xsc + d * c.Npad + qc];
    const double aq = lds_d[c.L.alpha + qc];
    const bool qv = qc < c.N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = BS * ba + rowof(r, c.q);
        double d2[D], r2 = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) { const double df = lds_d[c.L.xsc + d * c.Npad + p] - xq[d]; d2[d] = df * df; r2 += d2[d]; }
        double kf, gg;
        kfun<KN>(r2, kf, gg);
        double Q = kinv[r] - lds_d[c.L.alpha + p] * aq;
        Q = (qv && p < c.N) ? Q : 0.0;
        const double wq = wgt * Q;
        accsf = fma(wq, kf, accsf);
        const double wg = wq * gg;
#pragma unroll
        for (int d = 0; d < D; ++d) accl[d] = fma(wg, d2[d], accl[d]);
        if (p == qc) accsn += Q;
    }
}

// 16x16 diagonal block: W = L L^T, X = L^-1 by Gaussian elimination of [W | I] (lane i (and its mirrors in the other three
// 16-lane rows) owns row i), X = D^-1/2 L1^-1.  Out: S1 = X, S2 = X^T (acc layout), logsum, bad.
// The pivot row reaches the other rows as a DPP row broadcast (v_mov_b64_dpp row_newbcast:k -- lane k of every 16-lane row,
// the one 64-bit DPP control gfx950 has) straight into a VGPR: v_readlane pairs through one SGPR pair cost four
// instructions and a wait state per multiply-add (12.6 k cycles per block, the serial part (B) of phase_potrf 25 % of an
// evaluation of an N = 500 tile).  Same arithmetic in the same order: bit-identical results.
template <int K>
__device__ __forceinline__ double row_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true); }

template <int K>
__device__ __forceinline__ void diag_step(double (&a)[16], double (&e)[16], int g, double& mypiv, int& isbad) {
    double p = row_bcast<K>(a[K]);
    if (!(p > 0.0)) { isbad = 1; p = 1.0; }
    mypiv = (g == K) ? p : mypiv;
    // m = a[K] / p without v_div_scale / v_div_fmas / v_div_fixup (pivots are positive and far from the exponent range's
    // ends): v_rcp_f64 (24 bits) + one Newton step (48 bits), the quotient corrected by its residual -- 6 dependent
    // instructions instead of 12.  The chain shares its SIMD with an MFMA-dense wave of the CU's other workgroup, which
    // issues a 64-cycle MFMA into every dependence gap: depth, not count, is what this code costs.
    const double x0 = __builtin_amdgcn_rcp(p);
    const double x1 = fma(x0, fma(-p, x0, 1.0), x0);
    const double qq = a[K] * x1;
    double m = fma(fma(-p, qq, a[K]), x1, qq);
    m = (g > K) ? m : 0.0;
#pragma unroll
    for (int cc = K + 1; cc < 16; ++cc) a[cc] = fma(-m, row_bcast<K>(a[cc]), a[cc]);
#pragma unroll
    for (int cc = 0; cc <= K; ++cc) e[cc] = fma(-m, row_bcast<K>(e[cc]), e[cc]);
    if constexpr (K + 1 < 16) diag_step<K + 1>(a, e, g, mypiv, isbad);
}

__device__ __forceinline__ void diag_factor(const f64x4& W, int Ad, int lane, f64x4& S1, f64x4& S2, double& logsum, int& bad) {
    const int q = lane >> 4, g = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_d[Ad + rowof(r, q) * 17 + g] = W[r];
    wave_lds_sync();
    double a[16], e[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) { a[cc] = lds_d[Ad + g * 17 + cc]; e[cc] = (cc == g) ? 1.0 : 0.0; }
    int isbad = 0;
    double mypiv = 1.0;
    diag_step<0>(a, e, g, mypiv, isbad);
    const double rs = 1.0 / sqrt(mypiv);
    wave_lds_sync();
    if (q == 0) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) lds_d[Ad + g * 17 + cc] = e[cc] * rs;     // X row-major
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        S1[r] = lds_d[Ad + rowof(r, q) * 17 + g];
        S2[r] = lds_d[Ad + g * 17 + rowof(r, q)];
    }
    double lg = (lane < 16) ? 0.5 * log(mypiv) : 0.0;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) lg += __shfl_xor(lg, off);
    logsum = __shfl(lg, 0);
    bad = isbad;
}

// ---- phase 1: K = U^T U by panels of PR = 4 block rows, z = L^-1 y.
// HBM traffic sets the pace at N ~ 2000 (the 32 MB factor of a tile lives in HBM): with one block row per step every
// operand block U_ki was loaded once per PRODUCT (3.2 flop/B, measured 3.3 TB/s = HBM-bound at 22 % of the MFMA peak);
// with 4-row panels a wave keeps 4 x 2 accumulators and loads U_ki once per 4 products (the 4 panel blocks U_k,jr are
// common to all waves: L1/L2 hits).  Per panel: (A) the 10 diagonal-region sums K - sum_k U_k,jr^T U_k,jr' spread over
// the waves -> LDS; (B) wave 0 factorises the 4 x 4 block triangle (4 16x16 factorisations, 6 U_rr', forward solve);
// (C) every wave: column pairs right of the panel, k-loop then the in-panel triangular solve in registers.
constexpr int PR = 4;
__device__ __forceinline__ constexpr int pidx(int r, int r2) { return r * PR - (r * (r - 1)) / 2 + (r2 - r); }   // r <= r2 < 4

// ---- M = L^-1 (lower slots) and alpha = M^T z, the rows of ONE panel (i0 .. i0 + nr): called from the column phase of
// phase_potrf, right behind the panel's serial part -- everything it reads is in memory by then (earlier panels' rows of U
// and M, this panel's factors and in-panel U blocks, z) -- so that the inverse costs no barriers of its own and its items
// fill the column phase of the late panels, where few U columns are left.  A wave owns pairs of block columns, 4 x 2
// accumulators:   M_ij = -L_i^-1 sum_{k=j}^{i-1} U_ki^T M_kj   (M_jj = L_j^-1 already in the diagonal slot).
// alpha: one writer per column and panel, panels separated by (team) barriers, so every column's sum keeps its order; a team
// keeps alpha in its workspace.
template <int D, int KN, bool TEAM>
__device__ __forceinline__ void trtri_pair(Ctx<D, KN>& c, int i0, int nr, int p) {       // the column pair 2p, 2p + 1 < i0
    const int NB = c.NB, lane = c.lane;
    const int jc0 = 2 * p, jc1 = jc0 + 1;
    f64x4 acc[PR][2];
#pragma unroll
    for (int r = 0; r < PR; ++r) { acc[r][0] = zero4(); acc[r][1] = zero4(); }
    {
        f64x4 A[PR], B0, B1;
#pragma unroll
        for (int r = 0; r < PR; ++r) A[r] = ldg(c.ws, (r < nr) ? jc0 * NB + i0 + r : c.zb, lane);
        B0 = ldg(c.ws, jc0 * NB + jc0, lane);
        B1 = ldg(c.ws, c.zb, lane);                         // M_k,jc1 is zero for k = jc0 < jc1
        for (int k = jc0; k + 1 < i0; ++k) {          // last step peeled: unconditional loads, no copy-first
            f64x4 nA[PR], nB0, nB1;
#pragma unroll
            for (int r = 0; r < PR; ++r) nA[r] = ldg(c.ws, (r < nr) ? (k + 1) * NB + i0 + r : c.zb, lane);
            nB0 = ldg(c.ws, (k + 1) * NB + jc0, lane);
            nB1 = ldg(c.ws, (k + 1) * NB + jc1, lane);
#pragma unroll
            for (int r = 0; r < PR; ++r) { mma_blk(acc[r][0], A[r], B0); mma_blk(acc[r][1], A[r], B1); }
#pragma unroll
            for (int r = 0; r < PR; ++r) A[r] = nA[r];
            B0 = nB0; B1 = nB1;
        }
        if (jc0 < i0) {
#pragma unroll
            for (int r = 0; r < PR; ++r) { mma_blk(acc[r][0], A[r], B0); mma_blk(acc[r][1], A[r], B1); }
        }
    }
    double ap0 = 0.0, ap1 = 0.0;
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        if (r < nr) {
            const int ir = i0 + r;
            const f64x4 Lop = ldg(c.ws, c.dT0 + ir, lane);
            f64x4 X0 = zero4(), X1 = zero4();
            mma_blk(X0, Lop, acc[r][0]);
            mma_blk(X1, Lop, acc[r][1]);
            X0 = -X0; X1 = -X1;
            stg(c.ws, ir * NB + jc0, lane, X0);
            stg(c.ws, ir * NB + jc1, lane, X1);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const double zr = lds_d[c.L.z + BS * ir + rowof(rr, c.q)];
                ap0 = fma(X0[rr], zr, ap0);
                ap1 = fma(X1[rr], zr, ap1);
            }
#pragma unroll
            for (int r2 = r + 1; r2 < PR; ++r2) {
                if (r2 < nr) {
                    const f64x4 U = ldg(c.ws, ir * NB + i0 + r2, lane);
                    mma_blk(acc[r2][0], U, X0);
                    mma_blk(acc[r2][1], U, X1);
                }
            }
        }
    }
    ap0 = qsum(ap0); ap1 = qsum(ap1);
    if (c.q == 0) {                              // a column pair always belongs to the same wave: no race
        if (TEAM) {
            gdouble *a0 = c.ag + BS * jc0 + c.g, *a1 = c.ag + BS * jc1 + c.g;
            gst_d(a0, gld_d(a0) + ap0);
            gst_d(a1, gld_d(a1) + ap1);
        } else {
            lds_d[c.L.alpha + BS * jc0 + c.g] += ap0;
            lds_d[c.L.alpha + BS * jc1 + c.g] += ap1;
        }
    }
}

template <int D, int KN, bool TEAM>
__device__ __forceinline__ void trtri_triangle(Ctx<D, KN>& c, int i0, int nr) {           // the triangle inside the panel
    const int NB = c.NB, lane = c.lane;
    for (int j = i0; j < i0 + nr; ++j) {
        const f64x4 Mjj = ldg(c.ws, j * NB + j, lane);
        double ap = 0.0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ap = fma(Mjj[rr], lds_d[c.L.z + BS * j + rowof(rr, c.q)], ap);
        for (int i = j + 1; i < i0 + nr; ++i) {
            f64x4 acc = zero4();
            for (int k = j; k < i; ++k) {
                const f64x4 A = ldg(c.ws, k * NB + i, lane);        // U_ki
                const f64x4 B = ldg(c.ws, k * NB + j, lane);        // M_kj (k == j: diagonal slot)
                mma_blk(acc, A, B);
            }
            const f64x4 Lop = ldg(c.ws, c.dT0 + i, lane);
            f64x4 Mij = zero4();
            mma_blk(Mij, Lop, acc);
            Mij = -Mij;
            stg(c.ws, i * NB + j, lane, Mij);
            // this wave reads the block back a few instructions later (k = i of the next row): a write-through store
            // still on its way is not ordered before a load of the same bytes
            if (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) ap = fma(Mij[rr], lds_d[c.L.z + BS * i + rowof(rr, c.q)], ap);
        }
        const double a = qsum(ap);
        if (c.q == 0) {
            if (TEAM) { gdouble* aj = c.ag + BS * j + c.g; gst_d(aj, gld_d(aj) + a); }
            else lds_d[c.L.alpha + BS * j + c.g] += a;
        }
    }
}

// static deal (teams): pairs from the last wave down -- the first waves have the most U columns of the panel, wave 0 its
// serial part -- and the triangle on one wave
template <int D, int KN, bool TEAM>
__device__ __forceinline__ void trtri_panel(Ctx<D, KN>& c, int i0, int nr) {
    for (int p = c.nwt - 1 - c.vw; 2 * p < i0; p += c.nwt) trtri_pair<D, KN, TEAM>(c, i0, nr, p);
    if (c.vw == c.nwt - 1 - ((i0 >> 1) % c.nwt)) trtri_triangle<D, KN, TEAM>(c, i0, nr);
}

// acc += sum_{k in [kb, ke)} U_k,jb+r^T U_k,jb+r2 ;  tp += the lane's share of sum_k U_k,jb+r^T z_k (diagonal items).
// One product per step: the loop runs at the latency of its loads unless several steps are in flight.  A ring of PF operand
// pairs (the tail reloads the last pair, unused) keeps PF steps of loads outstanding; the products are issued in ascending k.
template <int D, int KN, bool TEAM>
__device__ __forceinline__ void diag_chain(const Ctx<D, KN>& c, int jb, int r, int r2, int kb, int ke, bool dg, f64x4& acc, double& tp) {
    if (kb >= ke) return;
    const int NB = c.NB, lane = c.lane;
    constexpr int PF = 6;
    f64x4 Ar[PF], Br[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int kk = min(kb + u, ke - 1);
        Ar[u] = ldg(c.ws, kk * NB + jb + r, lane);
        Br[u] = ldg(c.ws, kk * NB + jb + r2, lane);
    }
    for (int k0 = kb; k0 < ke; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k = k0 + u;
            if (k < ke) {
                mma_blk(acc, Ar[u], Br[u]);
                if (dg) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) tp = fma(Ar[u][rr], lds_d[c.L.z + BS * k + rowof(rr, c.q)], tp);
                }
                const int kn = min(k + PF, ke - 1);
                Ar[u] = ldg(c.ws, kn * NB + jb + r, lane);
                Br[u] = ldg(c.ws, kn * NB + jb + r2, lane);
            }
        }
    }
}

__device__ __forceinline__ void diag_item(int bb, int& r, int& r2) {
    r = 0; r2 = bb;
    if (bb >= 9) { r = 3; r2 = 3; } else if (bb >= 7) { r = 2; r2 = bb - 5; } else if (bb >= 4) { r = 1; r2 = bb - 3; }
}

// Look-ahead (one workgroup per tile): the diagonal region of the NEXT panel summed over the rows of all panels
// before the current one -- everything it needs is in memory when the current panel starts -- kept per lane in LDS (the
// accumulator and the partial t, exactly: the next panel continues the same sums with the current panel's four rows).  These ten
// chains are pulled by the waves that would otherwise wait for wave 0's serial part (B), and by every wave that runs out of
// columns in (C); what is left of (A) is four steps per item.
template <int D, int KN>
__device__ __forceinline__ void la_items(const Ctx<D, KN>& c, int j0, int jn, bool until_b_done) {
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
    const int nrn = min(PR, c.NB - jn);
    for (;;) {
        if (until_b_done && __hip_atomic_load(&sh->g0done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        int v = 0;
        if (c.lane == 0) v = atomicAdd(&sh->gnext[0], 1);
        const int bb = __builtin_amdgcn_readfirstlane(v);
        if (bb >= 10) break;
        int r, r2;
        diag_item(bb, r, r2);
        if (r2 >= nrn) continue;
        f64x4 acc = zero4();
        double tp = 0.0;
        diag_chain<D, KN, false>(c, jn, r, r2, 0, j0, r == r2, acc, tp);
        if (NW == 4) {                                   // 4-wave build: LDS; 8-wave build (large tiles fill the LDS): workspace
            stl(c.L.PnLA + bb * BLK, c.lane, acc);
            if (r == r2) lds_d[c.L.tpLA + 64 * r + c.lane] = tp;
        } else {
            stg_t<false>(c.ws, c.pn0 + bb, c.lane, acc);
            if (r == r2) c.ws[(size_t)(c.pn0 + 11) * BLK + 64 * r + c.lane] = tp;
        }
    }
}

// One item of the column phase (C): PCW block columns right of the panel j0 .. j0 + nr, k-loop over the rows above the panel,
// then the in-panel triangular solve with the panel's factors from LDS (pnb: the panel's ten blocks, slot (r,r) = (L_r^-1)^T,
// slot (r,r') = U_jr,jr').
template <int D, int KN, bool TEAM, int PCW>
__device__ __forceinline__ void cols_item(const Ctx<D, KN>& c, const int j0, const int nr, const int i0, const int pnb) {
    const int NB = c.NB, lane = c.lane;
    int ib[PCW];
#pragma unroll
    for (int cc = 0; cc < PCW; ++cc) ib[cc] = (i0 + cc < NB) ? i0 + cc : -1;
    f64x4 acc[PR][PCW];
#pragma unroll
    for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int cc = 0; cc < PCW; ++cc) acc[r][cc] = zero4();
    if (j0 > 0) {
        f64x4 A[PR], B[PCW];
#pragma unroll
        for (int r = 0; r < PR; ++r) A[r] = ldg(c.ws, (r < nr) ? j0 + r : c.zb, lane);
#pragma unroll
        for (int cc = 0; cc < PCW; ++cc) B[cc] = ldg(c.ws, ib[cc] >= 0 ? ib[cc] : c.zb, lane);
        // last step peeled: the loads of the next operands are unconditional (a conditional load makes the compiler
        // copy the whole operand set first)
        for (int k = 0; k + 1 < j0; ++k) {
            f64x4 nA[PR], nB[PCW];
#pragma unroll
            for (int r = 0; r < PR; ++r) nA[r] = ldg(c.ws, (r < nr) ? (k + 1) * NB + j0 + r : c.zb, lane);
#pragma unroll
            for (int cc = 0; cc < PCW; ++cc) nB[cc] = ldg(c.ws, ib[cc] >= 0 ? (k + 1) * NB + ib[cc] : c.zb, lane);
#pragma unroll
            for (int r = 0; r < PR; ++r)
#pragma unroll
                for (int cc = 0; cc < PCW; ++cc) mma_blk(acc[r][cc], A[r], B[cc]);
#pragma unroll
            for (int r = 0; r < PR; ++r) A[r] = nA[r];
#pragma unroll
            for (int cc = 0; cc < PCW; ++cc) B[cc] = nB[cc];
        }
#pragma unroll
        for (int r = 0; r < PR; ++r)
#pragma unroll
            for (int cc = 0; cc < PCW; ++cc) mma_blk(acc[r][cc], A[r], B[cc]);
    }
    // right-hand sides, then the in-panel triangular solve: X_r = L_r^-1 (W_r - sum_{r''<r} U_r''r^T X_r'')
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        if (r < nr) {
#pragma unroll
            for (int cc = 0; cc < PCW; ++cc)
                if (ib[cc] >= 0) acc[r][cc] = kblock<D, KN>(c, j0 + r, ib[cc]) - acc[r][cc];
        }
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        if (r < nr) {
            const f64x4 Lop = ldl(pnb + pidx(r, r) * BLK, lane);
            f64x4 X[PCW];
#pragma unroll
            for (int cc = 0; cc < PCW; ++cc) {
                X[cc] = zero4();
                mma_blk(X[cc], Lop, acc[r][cc]);
                if (ib[cc] >= 0) stg(c.ws, (j0 + r) * NB + ib[cc], lane, X[cc]);
            }
#pragma unroll
            for (int r2 = r + 1; r2 < PR; ++r2) {
                if (r2 < nr) {
                    const f64x4 U = ldl(pnb + pidx(r, r2) * BLK, lane);
#pragma unroll
                    for (int cc = 0; cc < PCW; ++cc) {
                        f64x4 T = zero4();
                        mma_blk(T, U, X[cc]);
                        acc[r2][cc] -= T;
                    }
                }
            }
        }
    }
}

template <int D, int KN, bool TEAM>
__device__ __forceinline__ void phase_potrf(Ctx<D, KN>& c, const bool want_m) {
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
    const int NB = c.NB, lane = c.lane, w = c.w;
    if (c.tid == 0) { sh->logdet = 0.0; sh->fail = 0; }
    if (want_m) {
        for (int idx = c.tid; idx < c.Npad; idx += NT) lds_d[c.L.alpha + idx] = 0.0;
        if (TEAM && c.member == 0)
            for (int idx = c.tid; idx < c.Npad; idx += NT) gst_d(c.ag + idx, 0.0);     // (drained by the first team barrier)
    }
    __syncthreads();
    unsigned long long tA = 0, tB = 0, tC = 0, tW = 0, t0 = 0;
    PROF_BEGIN();
    for (int j0 = 0; j0 < NB; j0 += PR) {
        const int nr = min(PR, NB - j0);
        if (TEAM) t0 = __builtin_amdgcn_s_memtime();
        // ---- (A) diagonal region: D_rr' = K_jr,jr' - sum_{k<j0} U_k,jr^T U_k,jr'  and  t_r = sum_{k<j0} U_k,jr^T z_k
        constexpr bool LA = !TEAM;                         // look-ahead of the next panel's diagonal region (la_items)
        if (LA && c.tid == 0) { sh->gnext[0] = (j0 + PR < NB) ? 0 : 10; sh->g0done = 0; sh->gnext[1] = 0; sh->hp[0] = 0; }
        for (int bb = c.vw; bb < 10; bb += c.nwt) {
            int r, r2;
            diag_item(bb, r, r2);
            if (r2 >= nr) continue;
            const bool dg = (r == r2);
            f64x4 acc = zero4();
            double tp = 0.0;
            if (LA && j0 > 0) {
                // rows of the panels before the previous one: summed ahead (la_items); the previous panel's rows now
                if (NW == 4) {
                    acc = ldl(c.L.PnLA + bb * BLK, lane);
                    if (dg) tp = lds_d[c.L.tpLA + 64 * r + lane];
                } else {
                    acc = ldg(c.ws, c.pn0 + bb, lane);
                    if (dg) tp = c.ws[(size_t)(c.pn0 + 11) * BLK + 64 * r + lane];
                }
                diag_chain<D, KN, TEAM>(c, j0, r, r2, j0 - PR, j0, dg, acc, tp);
            } else {
                diag_chain<D, KN, TEAM>(c, j0, r, r2, 0, j0, dg, acc, tp);
            }
            acc = kblock<D, KN>(c, j0 + r, j0 + r2) - acc;
            if (TEAM) stg(c.ws, c.pn0 + bb, lane, acc);
            else stl(c.L.Pn + bb * BLK, lane, acc);
            if (dg) {
                const double t = qsum(tp);
                if (c.q == 0) {
                    if (TEAM) gst_d(c.tpg + BS * r + c.g, t);
                    else lds_d[c.L.tp4 + BS * r + c.g] = t;
                }
            }
        }
        if (TEAM) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tW += t1 - t0; t0 = t1; }
        PROF_END(c, 0);
        team_barrier<TEAM>(c);
        PROF_END(c, 1);
        if (TEAM) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tA += t1 - t0; t0 = t1; }
        // ---- (B) the 4 x 4 block triangle (wave 0 of the owner): after it slot (r,r) of Pn holds (L_r^-1)^T, slot (r,r')
        // holds U_jr,jr'
        constexpr int PCW = TEAM ? 1 : (NW == 4 ? 2 : 3);
        // one workgroup: ONE queue over the panel's items of the column phase (C) -- the triangle of the inverse (the longest),
        // the U columns, the column pairs of the inverse (longest first)
        const int nT = want_m ? 1 : 0;
        const int nU = (NB - j0 - nr + PCW - 1) / PCW;
        auto queue_item = [&](const int idx) {
            if (idx < nT) { trtri_triangle<D, KN, TEAM>(c, j0, nr); PROF_END(c, 6); }
            else { cols_item<D, KN, TEAM, PCW>(c, j0, nr, j0 + nr + PCW * (idx - nT), c.L.Pn); PROF_END(c, 5); }
        };
        // The column pairs of the inverse are DEFERRED by one panel: the pairs of panel j0 - PR (everything they read has been in
        // memory since that panel's serial part; a pair only feeds the same pair of the next panel and the gradient phase) are
        // pulled from their own queue (sh->hp[0]) by the waves that would otherwise wait for wave 0's serial part (B) of THIS
        // panel -- 15 % of an evaluation of an N = 500 tile, after the look-ahead items had run out -- and what is left of them
        // closes the column phase.  The pairs of the last panel run after the loop.  Same arithmetic per pair and the same order
        // of a column's alpha terms (panel by panel): bit-identical results.
        auto deferred_pairs = [&](const int i0p, const int nrp, const bool until_b_done) {
            for (;;) {
                if (until_b_done && __hip_atomic_load(&sh->g0done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                int v = 0;
                if (lane == 0) v = atomicAdd(&sh->hp[0], 1);
                const int idx = __builtin_amdgcn_readfirstlane(v);
                if (2 * idx >= i0p) break;
                trtri_pair<D, KN, TEAM>(c, i0p, nrp, idx);            // ascending: longest k-loop first
                PROF_END(c, 7);
            }
        };
        if (LA && w != 0) {
            la_items<D, KN>(c, j0, j0 + PR, true);
            PROF_END(c, 3);
            if (want_m && j0 > 0) deferred_pairs(j0 - PR, PR, true);
        }
        if (w == 0 && c.member == 0) {
            // the serial part: this wave's VALU chain shares its SIMD with a wave of the CU's other workgroup, whose MFMAs
            // (64 cycles each, nothing else issues meanwhile) slip into every dependence gap -- measured 23 cycles per
            // instruction; with priority the arbiter takes this wave's instruction whenever one is ready
            __builtin_amdgcn_s_setprio(3);
            f64x4 Dd[10];
#pragma unroll
            for (int bb = 0; bb < 10; ++bb) Dd[bb] = TEAM ? ldg(c.ws, c.pn0 + bb, lane) : ldl(c.L.Pn + bb * BLK, lane);
            double tpx[PR] = {0.0, 0.0, 0.0, 0.0};      // per-lane partials of t_r' from the rows of this panel
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                if (r < nr) {
                    const int jr = j0 + r;
                    f64x4 S1, S2;
                    double ls;
                    int bad;
                    PROF_END(c, 2);
                    diag_factor(Dd[pidx(r, r)], c.L.Ad, lane, S1, S2, ls, bad);
                    PROF_END(c, 15);
                    stg(c.ws, jr * NB + jr, lane, S1);
                    stg(c.ws, c.dT0 + jr, lane, S2);
                    stl(c.L.Pn + pidx(r, r) * BLK, lane, S2);
                    if (TEAM) stg(c.ws, c.pn0 + pidx(r, r), lane, S2);
                    const double t = (TEAM ? gld_d(c.tpg + BS * r + c.g) : lds_d[c.L.tp4 + BS * r + c.g]) + qsum(tpx[r]);
                    if (c.q == 0) lds_d[c.L.tmp + c.g] = lds_d[c.L.y + BS * jr + c.g] - t;
                    wave_lds_sync();
                    double zz = 0.0;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) zz = fma(S2[rr], lds_d[c.L.tmp + rowof(rr, c.q)], zz);
                    zz = qsum(zz);
                    if (c.q == 0) {
                        lds_d[c.L.z + BS * jr + c.g] = zz;
                        if (TEAM) gst_d(c.zg + BS * jr + c.g, zz);
                    }
                    if (lane == 0) {
                        sh->logdet += ls;
                        if (bad) { sh->fail = 1; if (TEAM) __hip_atomic_store(&c.tc->fail, 1, RLX_AGENT); }
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int r2 = r + 1; r2 < PR; ++r2) {
                        if (r2 < nr) {
                            f64x4 U = zero4();
                            mma_blk(U, S2, Dd[pidx(r, r2)]);
                            Dd[pidx(r, r2)] = U;
                            stg(c.ws, jr * NB + j0 + r2, lane, U);
                            stl(c.L.Pn + pidx(r, r2) * BLK, lane, U);
                            if (TEAM) stg(c.ws, c.pn0 + pidx(r, r2), lane, U);
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) tpx[r2] = fma(U[rr], lds_d[c.L.z + BS * jr + rowof(rr, c.q)], tpx[r2]);
                        }
                    }
#pragma unroll
                    for (int r2 = r + 1; r2 < PR; ++r2) {
#pragma unroll
                        for (int r3 = r2; r3 < PR; ++r3) {
                            if (r3 < nr) {
                                f64x4 T = zero4();
                                mma_blk(T, Dd[pidx(r, r2)], Dd[pidx(r, r3)]);
                                Dd[pidx(r2, r3)] -= T;
                            }
                        }
                    }
                }
            }
            if (LA && lane == 0) __hip_atomic_store(&sh->g0done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_setprio(0);
            PROF_END(c, 2);
        }
        team_barrier<TEAM>(c);
        PROF_END(c, 4);
        if (TEAM) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tB += t1 - t0; t0 = t1; }
        if (TEAM) {
            // the owner's results of (B) for the other members: the panel's factors and U blocks into LDS, its rows of z, and
            // whether a pivot failed
            if (c.member != 0) {
                for (int bb = w; bb < 10; bb += NW) stl(c.L.Pn + bb * BLK, lane, ldg(c.ws, c.pn0 + bb, lane));
                for (int idx = c.tid; idx < nr * BS; idx += NT) lds_d[c.L.z + BS * j0 + idx] = gld_d(c.zg + BS * j0 + idx);
                if (c.tid == 0) sh->fail = __hip_atomic_load(&c.tc->fail, RLX_AGENT);
            }
            __syncthreads();
        }
        if (sh->fail) break;
        // ---- (C) columns right of the panel, PCW per wave and step: 4 x PCW accumulators, operands of step k+1 in flight.
        // The factor of an N = 2000 tile lives in HBM / Infinity Cache and this loop runs at what the fabric delivers (5.7 TB/s
        // measured with 4 x 2 accumulators = 0.75 block loads per product); 4 x 3 needs 7 loads per 12 products.
        // A team deals single columns (4 x 1 accumulators): its time per panel is the k-loop of ONE item (v_mfma_f64_16x16x4
        // takes 64 cycles: a 4 x 3 item runs 1.3 us per step however many workgroups there are), so more, thinner items
        // are what more workgroups can use.
        // The 4-wave build (small tiles, at most ~9 column triples per panel for 4 waves) deals pairs: the finer items
        // balance better (fp64 fit of N = 500 tiles +3.4 %; single columns: the same).
        if (TEAM) {
            // static deal over the team's waves
            for (int i0 = j0 + nr + PCW * c.vw; i0 < NB; i0 += PCW * c.nwt) cols_item<D, KN, TEAM, PCW>(c, j0, nr, i0, c.L.Pn);
            if (want_m) trtri_panel<D, KN, TEAM>(c, j0, nr);
        } else {
            // the waves finish together whatever the mix; every item is the same arithmetic whoever runs it, and alpha has one
            // writer per column and panel
            for (;;) {
                int v = 0;
                if (lane == 0) v = atomicAdd(&sh->gnext[1], 1);
                const int idx = __builtin_amdgcn_readfirstlane(v);
                if (idx >= nT + nU) break;
                queue_item(idx);
            }
            if (want_m && j0 > 0) deferred_pairs(j0 - PR, PR, false);
        }
        if (LA) la_items<D, KN>(c, j0, j0 + PR, false);
        PROF_END(c, 8);
        team_barrier<TEAM>(c);
        PROF_END(c, 9);
        if (TEAM) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); tC += t1 - t0; t0 = t1; }
    }
    if (!TEAM && want_m && !sh->fail) {              // the deferred pairs of the last panel (every wave; sh->fail is uniform here)
        const int j0l = ((NB - 1) / PR) * PR;
        if (c.tid == 0) sh->hp[0] = 0;
        __syncthreads();
        for (;;) {
            int v = 0;
            if (lane == 0) v = atomicAdd(&sh->hp[0], 1);
            const int idx = __builtin_amdgcn_readfirstlane(v);
            if (2 * idx >= j0l) break;
            trtri_pair<D, KN, TEAM>(c, j0l, NB - j0l, idx);
            PROF_END(c, 7);
        }
    }
    if (TEAM && c.tid == 0 && c.member == 0) {       // developer: 100 MHz ticks of the owner's thread 0 per part (GPSAT_DEBUG_TEAM_STATS)
        __hip_atomic_fetch_add(&c.tc->pad1[0], (int)tW, RLX_AGENT);
        __hip_atomic_fetch_add(&c.tc->pad1[1], (int)tA, RLX_AGENT);
        __hip_atomic_fetch_add(&c.tc->pad1[2], (int)tB, RLX_AGENT);
        __hip_atomic_fetch_add(&c.tc->pad1[3], (int)tC, RLX_AGENT);
    }
    if (TEAM && want_m && !sh->fail)
        for (int idx = c.tid; idx < c.Npad; idx += NT) lds_d[c.L.alpha + idx] = gld_d(c.ag + idx);
    __syncthreads();
}

// ---- phase 2: M = L^-1 (lower slots), alpha = M^T z, by panels of PR block rows (same blocking as phase_potrf:
// a wave owns pairs of block columns, 4 x 2 accumulators, every streamed block is loaded once per 4 products):
//   M_ij = -L_i^-1 sum_{k=j}^{i-1} U_ki^T M_kj   (M_jj = L_j^-1 already in the diagonal slot)
// ---- phase 3: K^-1 blocks (K^-1)_ab = sum_{c>=a} M_ca^T M_cb, a >= b, contracted with dK/dtheta: 4 block rows a x
// 2 block columns b per wave and step (the same 4 x 2 blocking; operands that would fall above the diagonal are the
// zero block)
// Every item (4 block rows x 2 block columns) leaves its D + 2 partial sums PER LANE in the workspace (gpart, aliasing the
// prediction scratch, idle during an evaluation); they are added in a fixed order by the owner -- the gradient does not
// depend on how the items were dealt over waves and workgroups.
template <int D, int KN, bool TEAM>
__device__ __forceinline__ void phase_grad(Ctx<D, KN>& c) {
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
    const int NB = c.NB, lane = c.lane;
    double* gpart = reinterpret_cast<double*>(reinterpret_cast<char*>(c.ws) + c.gp0);
    // one workgroup: the items (longest first: their k-loops run from a0 to NB) are pulled from a queue; a team deals them
    int mine = TEAM ? c.vw : -1;
    PROF_BEGIN();
    if (!TEAM) {
        if (c.tid == 0) sh->gradnext = 0;
        __syncthreads();
        int v = 0;
        if (lane == 0) v = atomicAdd(&sh->gradnext, 1);
        mine = __builtin_amdgcn_readfirstlane(v);
    }
    int item = 0;
    for (int a0 = 0; a0 < NB; a0 += PR) {
        const int na = min(PR, NB - a0);
        for (int b0 = 0; b0 < a0 + na; b0 += 2, ++item) {
            if (item != mine) continue;
            if (TEAM) {
                mine += c.nwt;
            } else {
                int v = 0;
                if (lane == 0) v = atomicAdd(&sh->gradnext, 1);
                mine = __builtin_amdgcn_readfirstlane(v);
            }
            const bool hb1 = b0 + 1 < NB;
            f64x4 acc[PR][2];
#pragma unroll
            for (int r = 0; r < PR; ++r) { acc[r][0] = zero4(); acc[r][1] = zero4(); }
            {
                // M_c,x is stored for c >= x only (c == x: diagonal slot); above the diagonal the slot holds U
                f64x4 A[PR], B0, B1;
#pragma unroll
                for (int r = 0; r < PR; ++r) A[r] = ldg(c.ws, (r < na && a0 >= a0 + r) ? a0 * NB + a0 + r : c.zb, lane);
                B0 = ldg(c.ws, (a0 >= b0) ? a0 * NB + b0 : c.zb, lane);
                B1 = ldg(c.ws, (hb1 && a0 >= b0 + 1) ? a0 * NB + b0 + 1 : c.zb, lane);
                for (int cc = a0; cc + 1 < NB; ++cc) {          // last step peeled: unconditional loads, no copy-first
                    f64x4 nA[PR], nB0, nB1;
                    const int cn = cc + 1;
#pragma unroll
                    for (int r = 0; r < PR; ++r) nA[r] = ldg(c.ws, (r < na && cn >= a0 + r) ? cn * NB + a0 + r : c.zb, lane);
                    nB0 = ldg(c.ws, (cn >= b0) ? cn * NB + b0 : c.zb, lane);
                    nB1 = ldg(c.ws, (hb1 && cn >= b0 + 1) ? cn * NB + b0 + 1 : c.zb, lane);
#pragma unroll
                    for (int r = 0; r < PR; ++r) { mma_blk(acc[r][0], A[r], B0); mma_blk(acc[r][1], A[r], B1); }
#pragma unroll
                    for (int r = 0; r < PR; ++r) A[r] = nA[r];
                    B0 = nB0; B1 = nB1;
                }
                if (a0 < NB) {
#pragma unroll
                    for (int r = 0; r < PR; ++r) { mma_blk(acc[r][0], A[r], B0); mma_blk(acc[r][1], A[r], B1); }
                }
            }
            PROF_END(c, 10);
            double accl[D];
#pragma unroll
            for (int d = 0; d < D; ++d) accl[d] = 0.0;
            double accsf = 0.0, accsn = 0.0;
#pragma unroll
            for (int r = 0; r < PR; ++r) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int a = a0 + r, b = b0 + n;
                    if (r < na && b < NB && a >= b)
                        contract<D, KN>(c, acc[r][n], a, b, (a == b) ? 1.0 : 2.0, accl, accsf, accsn);
                }
            }
            double* gp = gpart + (size_t)item * (D + 2) * 64 + lane;
            if (TEAM) {
                gdouble* gg = (gdouble*)gp;
#pragma unroll
                for (int d = 0; d < D; ++d) gst_d(gg + d * 64, accl[d]);
                gst_d(gg + D * 64, accsf);
                gst_d(gg + (D + 1) * 64, accsn);
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) gp[d * 64] = accl[d];
                gp[D * 64] = accsf;
                gp[(D + 1) * 64] = accsn;
            }
            PROF_END(c, 13);
        }
    }
    const int nitems = item;
    PROF_END(c, 10);
    team_barrier<TEAM>(c);
    PROF_END(c, 11);
    if (c.member != 0) return;
    // fixed-order sum: wave w adds the items w, w + NW, ... per lane, then across lanes, then across waves
    double v[D + 2];
#pragma unroll
    for (int i = 0; i < D + 2; ++i) v[i] = 0.0;
    for (int it = c.w; it < nitems; it += NW) {
        const double* gp = gpart + (size_t)it * (D + 2) * 64 + lane;
#pragma unroll
        for (int i = 0; i < D + 2; ++i) v[i] += TEAM ? gld_d((const gdouble*)(gp + i * 64)) : gp[i * 64];
    }
#pragma unroll
    for (int i = 0; i < D + 2; ++i) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D + 2; ++i) sh->red[c.w][i] = v[i];
    }
    __syncthreads();
    if (c.tid == 0) {
        for (int i = 0; i < D + 2; ++i) {
            double s = 0.0;
            for (int ww = 0; ww < NW; ++ww) s += sh->red[ww][i];
            if (i < D) sh->gth[i] = 0.5 * c.sf2 * s / sh->theta[i];
            else sh->gth[i] = 0.5 * s;
        }
    }
    __syncthreads();
}

template <int D, int KN, bool TEAM>
__device__ __forceinline__ void evaluate(Ctx<D, KN>& c, bool want_grad, const double* __restrict__ Xg) {
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
    __syncthreads();
    PROF_BEGIN();
    c.sf2 = sh->theta[D];
    c.sn2 = sh->theta[D + 1];
    for (int idx = c.tid; idx < c.Npad; idx += NT) {
#pragma unroll
        for (int d = 0; d < D; ++d) lds_d[c.L.xsc + d * c.Npad + idx] = (idx < c.N) ? Xg[(size_t)idx * D + d] / sh->theta[d] : 0.0;
    }
    __syncthreads();
    phase_potrf<D, KN, TEAM>(c, want_grad);
    if (sh->fail) {
        if (c.tid == 0) { sh->nll = __builtin_inf(); for (int i = 0; i < D + 2; ++i) sh->gth[i] = 0.0; }
        __syncthreads();
        return;
    }
    double qf = 0.0;
    for (int p = c.tid; p < c.N; p += NT) { const double zz = lds_d[c.L.z + p]; qf += zz * zz; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) qf += __shfl_xor(qf, off);
    if (c.lane == 0) sh->red[c.w][7] = qf;
    __syncthreads();
    if (c.tid == 0) {
        double s = 0.0;
        for (int ww = 0; ww < NW; ++ww) s += sh->red[ww][7];
        sh->nll = 0.5 * s + sh->logdet + 0.5 * (double)c.N * 1.8378770664093453;
    }
    __syncthreads();
    if (want_grad) phase_grad<D, KN, TEAM>(c);
    if (c.tid == 0) {
        sh->n_eval += 1;
        if (!(sh->nll == sh->nll)) sh->fail = 1;
    }
    __syncthreads();
    PROF_END(c, 12);
}

template <int D, int KN, bool TEAM>
__device__ __forceinline__ void predict_tile(Ctx<D, KN>& c, const double* __restrict__ Xs, double* __restrict__ fm,
                                             double* __restrict__ fv, double* __restrict__ yv, const double* theta,
                                             double* __restrict__ fcov) {
    const int NB = c.NB, lane = c.lane;
    const int PC = (c.P + BS - 1) / BS;
    // V = L^-1 K_* for two 16-column chunks per wave, by panels of PR block rows like phase_potrf: 4 x 2 accumulators,
    // every V block (wave-private scratch) is loaded once per 4 products and the factor once per pair of chunks
    for (int pc0 = 2 * c.w; pc0 < PC; pc0 += 2 * NW) {
        // this wave's V scratch [2 chunks][NB] blocks, or (full covariance wanted) the per-tile store of all chunks
        const int v0 = fcov ? c.cv0 + pc0 * NB : c.vs0 + c.w * 4 * NB;
        double xa[2][D];
        bool va[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int qa = BS * (pc0 + n) + c.g;
            va[n] = qa < c.P;
#pragma unroll
            for (int d = 0; d < D; ++d) xa[n][d] = va[n] ? Xs[(size_t)qa * D + d] / theta[d] : 0.0;
        }
        double vs[2] = {0.0, 0.0}, ms[2] = {0.0, 0.0};
        for (int j0 = 0; j0 < NB; j0 += PR) {
            const int nr = min(PR, NB - j0);
            f64x4 acc[PR][2];
#pragma unroll
            for (int r = 0; r < PR; ++r) { acc[r][0] = zero4(); acc[r][1] = zero4(); }
            if (j0 > 0) {
                f64x4 A[PR], B0, B1;
#pragma unroll
                for (int r = 0; r < PR; ++r) A[r] = ldg(c.ws, (r < nr) ? j0 + r : c.zb, lane);
                B0 = ldg(c.ws, v0, lane);
                B1 = ldg(c.ws, v0 + NB, lane);
                for (int k = 0; k + 1 < j0; ++k) {          // last step peeled: unconditional loads, no copy-first
                    f64x4 nA[PR], nB0, nB1;
#pragma unroll
                    for (int r = 0; r < PR; ++r) nA[r] = ldg(c.ws, (r < nr) ? (k + 1) * NB + j0 + r : c.zb, lane);
                    nB0 = ldg(c.ws, v0 + k + 1, lane);
                    nB1 = ldg(c.ws, v0 + NB + k + 1, lane);
#pragma unroll
                    for (int r = 0; r < PR; ++r) { mma_blk(acc[r][0], A[r], B0); mma_blk(acc[r][1], A[r], B1); }
#pragma unroll
                    for (int r = 0; r < PR; ++r) A[r] = nA[r];
                    B0 = nB0; B1 = nB1;
                }
                if (0 < j0) {
#pragma unroll
                    for (int r = 0; r < PR; ++r) { mma_blk(acc[r][0], A[r], B0); mma_blk(acc[r][1], A[r], B1); }
                }
            }
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                if (r < nr) {
                    const int jr = j0 + r;
                    const f64x4 Lop = ldg(c.ws, c.dT0 + jr, lane);
                    f64x4 V[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f64x4 Wb = ksblock<D, KN>(c, jr, xa[n], va[n]) - acc[r][n];
                        V[n] = zero4();
                        mma_blk(V[n], Lop, Wb);
                        stg(c.ws, v0 + n * NB + jr, lane, V[n]);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            vs[n] = fma(V[n][rr], V[n][rr], vs[n]);
                            ms[n] = fma(V[n][rr], lds_d[c.L.z + BS * jr + rowof(rr, c.q)], ms[n]);
                        }
                    }
#pragma unroll
                    for (int r2 = r + 1; r2 < PR; ++r2) {
                        if (r2 < nr) {
                            const f64x4 U = ldg(c.ws, jr * NB + j0 + r2, lane);
                            mma_blk(acc[r2][0], U, V[0]);
                            mma_blk(acc[r2][1], U, V[1]);
                        }
                    }
                }
            }
            if (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the V blocks stored above are this wave's next operands
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const double vsum = qsum(vs[n]), msum = qsum(ms[n]);
            const int qa = BS * (pc0 + n) + c.g;
            if (c.q == 0 && va[n]) {
                const double var = c.sf2 - vsum;
                fm[qa] = msum; fv[qa] = var; yv[qa] = var + c.sn2;
            }
        }
    }
    if (fcov) {
        // f*_cov = K_** - V^T V by 16 x 16 blocks (p <= q, mirrored), gpflow_models.py:245-263 (predict_f full_cov)
        if (TEAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // write-through V blocks of the other waves
        __syncthreads();
        int idx = 0;
        for (int p = 0; p < PC; ++p) {
            for (int q = p; q < PC; ++q, ++idx) {
                if ((idx & (NW - 1)) != c.w) continue;
                f64x4 Cb = zero4();
                f64x4 A = ldg(c.ws, c.cv0 + p * NB, lane), B = ldg(c.ws, c.cv0 + q * NB, lane);
                for (int k = 0; k < NB; ++k) {
                    f64x4 nA = A, nB = B;
                    if (k + 1 < NB) {
                        nA = ldg(c.ws, c.cv0 + p * NB + k + 1, lane);
                        nB = ldg(c.ws, c.cv0 + q * NB + k + 1, lane);
                    }
                    mma_blk(Cb, A, B);
                    A = nA; B = nB;
                }
                const int qj = BS * q + c.g;
                const bool vj = qj < c.P;
                double xq[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xq[d] = vj ? Xs[(size_t)qj * D + d] / theta[d] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pi = BS * p + rowof(r, c.q);
                    if (vj && pi < c.P) {
                        double r2 = 0.0;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            const double df = Xs[(size_t)pi * D + d] / theta[d] - xq[d];
                            r2 = fma(df, df, r2);
                        }
                        double kf, gg;
                        kfun<KN>(r2, kf, gg);
                        const double v = c.sf2 * kf - Cb[r];
                        fcov[(size_t)pi * c.P + qj] = v;
                        if (p != q) fcov[(size_t)qj * c.P + pi] = v;
                    }
                }
            }
        }
    }
}

template <int D, int KN>
__global__ void __launch_bounds__(NT, F64_MIN_WG) gp_tile_kernel_f64(const KernelArgs A) {
    constexpr int H = D + 2;
    Ctx<D, KN> c;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.w = c.tid >> 6;
    c.q = c.lane >> 4;
    c.g = c.lane & 15;
    const int NPmax = A.NBmax * BS;
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
#ifdef GPSAT_PROFILE
    if (threadIdx.x < NW * 16) sh->prof[threadIdx.x] = 0ull;
#endif
    int off = (int)((sizeof(Shared) + 15) / 16) * 2;
    c.L.xsc = off; off += D * NPmax;
    c.L.y = off; off += NPmax;
    c.L.z = off; off += NPmax;
    c.L.alpha = off; off += NPmax;
    c.L.LT = off; off += BLK;
    c.L.Ad = off; off += 16 * 17;
    c.L.tmp = off; off += 16;
    c.L.Pn = off; off += 10 * BLK;         // diagonal region of the current panel (phase_potrf)
    c.L.tp4 = off; off += 4 * BS;
    c.L.PnLA = 0; c.L.tpLA = 0;
#ifdef GPSAT_F64_W4
    c.L.PnLA = off; off += 10 * BLK;       // look-ahead sums of the next panel's diagonal region (la_items)
    c.L.tpLA = off; off += 4 * 64;
#endif
    double* wsall = reinterpret_cast<double*>(A.ws);
    const size_t stride = A.ws_stride;                 // doubles per workgroup
    c.ws = wsall + (size_t)blockIdx.x * stride;
    c.zb = (int)(stride / BLK) - 1;
    for (int i = c.tid; i < BLK; i += NT) c.ws[(size_t)c.zb * BLK + i] = 0.0;
    c.vw = c.w; c.nwt = NW; c.member = 0; c.G = 1; c.tc = nullptr; c.tpg = nullptr; c.zg = nullptr; c.ag = nullptr;
    c.pn0 = c.zb - 40;           // the exchange area of teams doubles as the look-ahead store of the 8-wave build
    const double* X = reinterpret_cast<const double*>(A.X);
    const double* y = reinterpret_cast<const double*>(A.y);
    const double* Xs = reinterpret_cast<const double*>(A.Xs);
    double* f_mean = reinterpret_cast<double*>(A.f_mean);
    double* f_var = reinterpret_cast<double*>(A.f_var);
    double* f_cov = reinterpret_cast<double*>(A.f_cov);
    double* y_var = reinterpret_cast<double*>(A.y_var);
    OptCfg o;
    o.optimiser = A.optimiser; o.max_iter = A.max_iter; o.max_ls = A.max_ls; o.want_grad_out = A.grad != nullptr;
    o.ftol = A.ftol; o.gtol = A.gtol; o.adam_lr = A.adam_lr; o.noise_rel = A.noise_rel;

    const bool sliced = A.seg_cost > 0;          // time-sliced tile queue, as in the fp32 kernels (gpsat_kernels.hip)
    for (;;) {
        __syncthreads();
        if (c.tid == 0) {
            if (sliced) {
                sh->tile = ring_pop(A);
            } else {
                const int slot = atomicAdd(A.queue, 1);
                sh->tile = slot < A.T ? A.order[slot] : -1;
            }
        }
        __syncthreads();
        const int entry = sh->tile;
        if (entry == -1) break;
        const int t = entry & 0x7fffffff;
        const bool resumed = entry < 0;
        const long long o0 = A.obs_off[t], o1 = A.obs_off[t + 1];
        const long long p0 = A.pred_off[t], p1 = A.pred_off[t + 1];
        c.N = (int)(o1 - o0);
        c.P = (int)(p1 - p0);
        c.NB = (c.N + BS - 1) / BS;
        c.Npad = c.NB * BS;
        const int NB = c.NB;
        c.dT0 = NB * NB;
        c.vs0 = c.dT0 + NB;
        c.cv0 = c.vs0 + NW * 4 * NB;
        c.gp0 = c.vs0 * (BLK * 8);
        if (c.N == 0) {
            if (c.tid == 0) {
                A.status[t] = 4; A.n_eval[t] = 0; A.nll[t] = 0.0;
                if (A.n_iter) A.n_iter[t] = 0;
                for (int i = 0; i < H; ++i) {
                    A.theta[(size_t)t * H + i] = A.theta0[(size_t)t * H + i];
                    if (A.grad) A.grad[(size_t)t * H + i] = 0.0;
                }
            }
            for (long long qq = p0 + c.tid; qq < p1; qq += NT) {
                const double sf2 = A.theta0[(size_t)t * H + D], sn2 = A.theta0[(size_t)t * H + D + 1];
                f_mean[qq] = 0.0; f_var[qq] = sf2; y_var[qq] = sf2 + sn2;
            }
            if (f_cov) {
                // prior covariance K_** of an empty tile
                const int Pn = (int)(p1 - p0);
                const double sf2 = A.theta0[(size_t)t * H + D];
                for (long long e = c.tid; e < (long long)Pn * Pn; e += NT) {
                    const int i = (int)(e / Pn), j = (int)(e % Pn);
                    double r2 = 0.0;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const double df = (Xs[(size_t)(p0 + i) * D + d] - Xs[(size_t)(p0 + j) * D + d]) / A.theta0[(size_t)t * H + d];
                        r2 = fma(df, df, r2);
                    }
                    double kf, gg;
                    kfun<KN>(r2, kf, gg);
                    f_cov[A.cov_off[t] + e] = sf2 * kf;
                }
            }
            if (sliced && c.tid == 0) __hip_atomic_fetch_add(&A.ring_ctl[32], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        for (int idx = c.tid; idx < c.Npad; idx += NT) {
            lds_d[c.L.y + idx] = (idx < c.N) ? y[o0 + idx] : 0.0;
            lds_d[c.L.z + idx] = 0.0;
            lds_d[c.L.alpha + idx] = 0.0;
        }
        if (resumed) {
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
                sh->shift[i] = (!box && i == D + 1) ? 1e-6 : 0.0;
                sh->trainable[i] = A.trainable[i] ? 1 : 0;
                sh->theta[i] = A.theta0[(size_t)t * H + i];
                sh->u[i] = u_of_theta(sh, i, sh->theta[i]);
                sh->m1[i] = 0.0; sh->m2[i] = 0.0;
            }
            const bool optim = (o.optimiser != 0 && o.max_iter > 0);
            sh->phase = optim ? PH_INIT : PH_FINAL;
            sh->want_grad = optim ? 1 : o.want_grad_out;
        }
        __syncthreads();
        const int seg_evals = sliced ? max(1, A.seg_cost / (NB * NB * NB)) : 0x7fffffff;
        bool suspended = false;
        for (int nseg = 1;; ++nseg) {
            evaluate<D, KN, false>(c, sh->want_grad != 0, X + (size_t)o0 * D);
            if (c.tid == 0) opt_advance(sh, H, o);
            __syncthreads();
            if (sh->phase == PH_EXIT) break;
            if (nseg >= seg_evals && sh->phase != PH_FINAL) { suspended = true; break; }
        }
        if (suspended) {
            unsigned* dst = A.state + (size_t)t * A.state_words;
            const unsigned* src = reinterpret_cast<const unsigned*>(sh);
            for (int i = c.tid; i < A.state_words; i += NT)
                __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every storing wave drains its own stores, then the barrier, then one lane publishes (see gpsat_kernels.hip)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c.tid == 0) ring_push(A, t);
            continue;
        }
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
                predict_tile<D, KN, false>(c, Xs + (size_t)p0 * D, f_mean + p0, f_var + p0, y_var + p0, sh->theta,
                                    f_cov ? f_cov + A.cov_off[t] : nullptr);
            } else {
                for (long long qq = p0 + c.tid; qq < p1; qq += NT) {
                    f_mean[qq] = __builtin_nan(""); f_var[qq] = __builtin_nan(""); y_var[qq] = __builtin_nan("");
                }
                if (f_cov)
                    for (long long qq = A.cov_off[t] + c.tid; qq < A.cov_off[t + 1]; qq += NT) f_cov[qq] = __builtin_nan("");
            }
        }
        if (sliced && c.tid == 0) __hip_atomic_fetch_add(&A.ring_ctl[32], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef GPSAT_PROFILE
    __syncthreads();
    if (A.prof && c.tid < NW * 16) atomicAdd(&A.prof[c.tid], sh->prof[c.tid]);
#endif
}

#ifndef GPSAT_F64_W4
// ---------------------------------------------------------------------------------------------
// the team kernel: KernelArgs::team_size workgroups per tile (see "Teams" above).  Workgroup b is member b % G of team
// b / G; the team's workspace is the slab of its member 0.  The owner pops tiles, runs the optimiser and, before every
// evaluation, publishes parameters and a command in the team's control block; everybody meets at the command barrier.
// ---------------------------------------------------------------------------------------------
template <int D, int KN>
__global__ void __launch_bounds__(NT, 1) gp_team_kernel_f64(const KernelArgs A) {
    constexpr int H = D + 2;
    constexpr bool TEAM = true;
    Ctx<D, KN> c;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.w = c.tid >> 6;
    c.q = c.lane >> 4;
    c.g = c.lane & 15;
    const int NPmax = A.NBmax * BS;
    Shared* sh = reinterpret_cast<Shared*>(lds_d);
    int off = (int)((sizeof(Shared) + 15) / 16) * 2;
    c.L.xsc = off; off += D * NPmax;
    c.L.y = off; off += NPmax;
    c.L.z = off; off += NPmax;
    c.L.alpha = off; off += NPmax;
    c.L.LT = off; off += BLK;
    c.L.Ad = off; off += 16 * 17;
    c.L.tmp = off; off += 16;
    c.L.Pn = off; off += 10 * BLK;
    c.L.tp4 = off; off += 4 * BS;
    c.L.PnLA = 0; c.L.tpLA = 0;
    const int G = A.team_size;
    c.G = G;
    c.member = (int)blockIdx.x % G;
    const int team = (int)blockIdx.x / G;
    c.vw = c.member * NW + c.w;
    c.nwt = NW * G;
    c.tc = (gTeamCtl*)A.team_ctl + team;
    const size_t stride = A.ws_stride;
    c.ws = reinterpret_cast<double*>(A.ws) + (size_t)team * stride;               // one slab per team
    c.zb = (int)(stride / BLK) - 1;
    c.pn0 = c.zb - 40;
    c.tpg = (gdouble*)(c.ws + (size_t)(c.zb - 29) * BLK);
    c.zg = (gdouble*)(c.ws + (size_t)(c.zb - 28) * BLK);
    c.ag = (gdouble*)(c.ws + (size_t)(c.zb - 14) * BLK);
    if (c.member == 0 && c.w == 0) { const f64x4 z0 = zero4(); stg(c.ws, c.zb, c.lane, z0); }
    if (c.tid == 0) sh->hp[0] = 0;                      // team barriers passed
    const double* X = reinterpret_cast<const double*>(A.X);
    const double* y = reinterpret_cast<const double*>(A.y);
    const double* Xs = reinterpret_cast<const double*>(A.Xs);
    double* f_mean = reinterpret_cast<double*>(A.f_mean);
    double* f_var = reinterpret_cast<double*>(A.f_var);
    double* f_cov = reinterpret_cast<double*>(A.f_cov);
    double* y_var = reinterpret_cast<double*>(A.y_var);
    OptCfg o;
    o.optimiser = A.optimiser; o.max_iter = A.max_iter; o.max_ls = A.max_ls; o.want_grad_out = A.grad != nullptr;
    o.ftol = A.ftol; o.gtol = A.gtol; o.adam_lr = A.adam_lr; o.noise_rel = A.noise_rel;
    __syncthreads();

    auto set_tile = [&](int t) {
        const long long o0 = A.obs_off[t], o1 = A.obs_off[t + 1];
        c.N = (int)(o1 - o0);
        c.P = (int)(A.pred_off[t + 1] - A.pred_off[t]);
        c.NB = (c.N + BS - 1) / BS;
        c.Npad = c.NB * BS;
        c.dT0 = c.NB * c.NB;
        c.vs0 = c.dT0 + c.NB;
        c.cv0 = c.vs0 + NW * 4 * c.NB;
        c.gp0 = c.vs0 * (BLK * 8);
    };

    if (c.member != 0) {
        // ---- a member: follow the owner's commands
        int t = -1;
        for (;;) {
            team_barrier<TEAM>(c, true);                                   // the command barrier
            if (c.tid == 0) {
                sh->hp[1] = __hip_atomic_load(&c.tc->cmd, RLX_AGENT);
                sh->hp[2] = __hip_atomic_load(&c.tc->tile, RLX_AGENT);
                sh->hp[3] = __hip_atomic_load(&c.tc->want_grad, RLX_AGENT);
                if (__hip_atomic_load(&c.tc->timeout, RLX_AGENT)) sh->hp[1] = TEAM_EXIT;
                for (int i = 0; i < H; ++i) sh->theta[i] = gld_d(&c.tc->theta[i]);
            }
            __syncthreads();
            const int cmd = sh->hp[1];
            if (cmd == TEAM_EXIT) break;
            if (cmd == TEAM_TILE) { t = sh->hp[2]; set_tile(t); continue; }
            evaluate<D, KN, TEAM>(c, sh->hp[3] != 0, X + (size_t)A.obs_off[t] * D);
        }
        return;
    }

    // ---- the owner
    auto command = [&](int cmd, int t, int want_grad) {
        if (c.tid == 0) {
            __hip_atomic_store(&c.tc->tile, t, RLX_AGENT);
            __hip_atomic_store(&c.tc->want_grad, want_grad, RLX_AGENT);
            for (int i = 0; i < H; ++i) gst_d(&c.tc->theta[i], sh->theta[i]);
            __hip_atomic_store(&c.tc->fail, 0, RLX_AGENT);
            __hip_atomic_store(&c.tc->cmd, cmd, RLX_AGENT);
        }
        team_barrier<TEAM>(c, true);
    };
    for (;;) {
        __syncthreads();
        if (c.tid == 0) {
            const int slot = atomicAdd(A.queue, 1);
            sh->tile = slot < A.T ? A.order[slot] : -1;
            if (__hip_atomic_load(&c.tc->timeout, RLX_AGENT)) sh->tile = -1;
        }
        __syncthreads();
        const int t = sh->tile;
        if (t == -1) { command(TEAM_EXIT, 0, 0); break; }
        const long long o0 = A.obs_off[t];
        const long long p0 = A.pred_off[t], p1 = A.pred_off[t + 1];
        set_tile(t);
        if (c.N == 0) {
            if (c.tid == 0) {
                A.status[t] = 4; A.n_eval[t] = 0; A.nll[t] = 0.0;
                if (A.n_iter) A.n_iter[t] = 0;
                for (int i = 0; i < H; ++i) {
                    A.theta[(size_t)t * H + i] = A.theta0[(size_t)t * H + i];
                    if (A.grad) A.grad[(size_t)t * H + i] = 0.0;
                }
            }
            for (long long qq = p0 + c.tid; qq < p1; qq += NT) {
                const double sf2 = A.theta0[(size_t)t * H + D], sn2 = A.theta0[(size_t)t * H + D + 1];
                f_mean[qq] = 0.0; f_var[qq] = sf2; y_var[qq] = sf2 + sn2;
            }
            if (f_cov)
                for (long long qq = A.cov_off[t] + c.tid; qq < A.cov_off[t + 1]; qq += NT) f_cov[qq] = __builtin_nan("");
            continue;
        }
        for (int idx = c.tid; idx < c.Npad; idx += NT) {
            lds_d[c.L.y + idx] = (idx < c.N) ? y[o0 + idx] : 0.0;
            lds_d[c.L.z + idx] = 0.0;
            lds_d[c.L.alpha + idx] = 0.0;
        }
        if (c.tid == 0) {
            sh->n_eval = 0; sh->n_eval_opt = 0; sh->status = 5; sh->iter = 0; sh->hist_n = 0; sh->hist_pos = 0;
            sh->last_dec = 1e300;
            sh->fail = 0;
            for (int i = 0; i < H; ++i) {
                const double lo = A.lo[(size_t)t * H + i], hi = A.hi[(size_t)t * H + i];
                const bool box = (lo == lo) && (hi == hi) && (fabs(lo) < 1e300) && (fabs(hi) < 1e300);
                sh->box[i] = box ? 1 : 0;
                sh->lo[i] = lo; sh->hi[i] = hi;
                sh->shift[i] = (!box && i == D + 1) ? 1e-6 : 0.0;
                sh->trainable[i] = A.trainable[i] ? 1 : 0;
                sh->theta[i] = A.theta0[(size_t)t * H + i];
                sh->u[i] = u_of_theta(sh, i, sh->theta[i]);
                sh->m1[i] = 0.0; sh->m2[i] = 0.0;
            }
            const bool optim = (o.optimiser != 0 && o.max_iter > 0);
            sh->phase = optim ? PH_INIT : PH_FINAL;
            sh->want_grad = optim ? 1 : o.want_grad_out;
        }
        __syncthreads();
        command(TEAM_TILE, t, 0);
        for (;;) {
            command(TEAM_EVAL, t, sh->want_grad);
            evaluate<D, KN, TEAM>(c, sh->want_grad != 0, X + (size_t)o0 * D);
            if (c.tid == 0) {
                if (__hip_atomic_load(&c.tc->timeout, RLX_AGENT)) sh->fail = 1;       // a barrier gave up: nothing of this is valid
                opt_advance(sh, H, o);
                if (__hip_atomic_load(&c.tc->timeout, RLX_AGENT)) sh->phase = PH_EXIT;
            }
            __syncthreads();
            if (sh->phase == PH_EXIT) break;
        }
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
                predict_tile<D, KN, TEAM>(c, Xs + (size_t)p0 * D, f_mean + p0, f_var + p0, y_var + p0, sh->theta,
                                          f_cov ? f_cov + A.cov_off[t] : nullptr);
            } else {
                for (long long qq = p0 + c.tid; qq < p1; qq += NT) {
                    f_mean[qq] = __builtin_nan(""); f_var[qq] = __builtin_nan(""); y_var[qq] = __builtin_nan("");
                }
                if (f_cov)
                    for (long long qq = A.cov_off[t] + c.tid; qq < A.cov_off[t + 1]; qq += NT) f_cov[qq] = __builtin_nan("");
            }
        }
    }
}
#endif

template <int D, int KN>
static hipError_t launch_one(const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
#ifndef GPSAT_F64_W4
    if (a.team_size > 1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gp_team_kernel_f64<D, KN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gp_team_kernel_f64<D, KN>), dim3(grid), dim3(NT), smem, stream, a);
        return hipGetLastError();
    }
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gp_tile_kernel_f64<D, KN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gp_tile_kernel_f64<D, KN>), dim3(grid), dim3(NT), smem, stream, a);
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

}  // namespace F64NS

size_t F64FN(shared_bytes_f64)(int D, int NBmax) {
    const size_t NP = (size_t)NBmax * F64NS::BS;
    size_t dbl = (sizeof(F64NS::Shared) + 15) / 16 * 2 + D * NP + 3 * NP + F64NS::BLK + 16 * 17 + 16 + 10 * F64NS::BLK + 4 * F64NS::BS + 2;
#ifdef GPSAT_F64_W4
    dbl += 10 * F64NS::BLK + 4 * 64;       // look-ahead sums (la_items)
#endif
    return (dbl * sizeof(double) + 15) & ~size_t(15);
}

int F64FN(state_words_f64)() { return (int)((sizeof(F64NS::Shared) + 15) / 16) * 4; }

size_t F64FN(workspace_doubles_per_wg_f64)(int NBmax, int PCcov) {
    // + V of all prediction chunks when the full covariance is wanted (spare chunks: a wave always solves 2 at a time)
    const size_t cov = PCcov > 0 ? (size_t)(PCcov + 3) * NBmax : 0;
    // ... 40 blocks of exchange areas for teams (diagonal region of a panel, z, alpha), and the zero block
    return (size_t)F64NS::BLK * ((size_t)NBmax * NBmax + (size_t)NBmax + (size_t)F64NS::NW * 4 * NBmax + cov + 40 + 1);
}

hipError_t F64FN(launch_tiles_f64)(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
    switch (D) {
        case 1: return F64NS::launch_d<1>(a, grid, smem, stream);
        case 2: return F64NS::launch_d<2>(a, grid, smem, stream);
        case 3: return F64NS::launch_d<3>(a, grid, smem, stream);
        case 4: return F64NS::launch_d<4>(a, grid, smem, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gpsat
