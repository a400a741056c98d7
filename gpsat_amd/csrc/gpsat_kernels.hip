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

namespace gpsat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------
// block movement (acc layout) and the MFMA chain
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 load_blk(const float* __restrict__ p, int lane) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p) + lane;
    f32x4 a = q[0], b = q[64], c = q[128], d = q[192];
    f32x16 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    r[8] = c[0]; r[9] = c[1]; r[10] = c[2]; r[11] = c[3];
    r[12] = d[0]; r[13] = d[1]; r[14] = d[2]; r[15] = d[3];
    return r;
}

__device__ __forceinline__ void store_blk(float* __restrict__ p, int lane, const f32x16& v) {
    f32x4* q = reinterpret_cast<f32x4*>(p) + lane;
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    f32x4 c = {v[8], v[9], v[10], v[11]}, d = {v[12], v[13], v[14], v[15]};
    q[0] = a; q[64] = b; q[128] = c; q[192] = d;
}

// acc += S_A^T * S_B   (16 x v_mfma_f32_32x32x2_f32)
__device__ __forceinline__ void mma_blk(f32x16& acc, const f32x16& a, const f32x16& b) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
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

// ---------------------------------------------------------------------------------------------
// covariance functions.  r2 is the squared scaled distance.
//   kf = k(r) / 1 , gg = g(r) with dk/dl_d = g(r) (x_d-x'_d)^2 / l_d^3   (both without sigma_f^2)
// ---------------------------------------------------------------------------------------------
template <int KERN>
__device__ __forceinline__ void kfun(float r2, float& kf, float& gg) {
    if (KERN == 0) {                       // RBF: exp(-r2/2)
        kf = __expf(-0.5f * r2);
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

// ---------------------------------------------------------------------------------------------
// per-workgroup context
// ---------------------------------------------------------------------------------------------
constexpr int NW = 4;          // waves per workgroup
constexpr int NT = 256;        // threads per workgroup
constexpr int BLK = 1024;      // floats per block
constexpr int HMAX = 6;        // max D + 2 (D <= 4)
constexpr int MH = 8;          // L-BFGS history

struct Shared {
    // evaluation interface
    double theta[HMAX];
    double gth[HMAX];          // dNLL/dtheta
    double nll;
    double logdet;
    double red[NW][8];
    // optimiser state (thread 0 writes, everybody reads after a barrier)
    double lo[HMAX], hi[HMAX], shift[HMAX];
    double u[HMAX], g[HMAX], f;            // current accepted point (u-space)
    double ut[HMAX], gt[HMAX], ft;         // trial point
    double d[HMAX];
    double S[MH][HMAX], Y[MH][HMAX], rho_[MH];
    double m1[HMAX], m2[HMAX];             // Adam moments
    // line search
    double t, t_prev, f_prev, dphi_prev, t_lo, f_lo, dphi_lo, t_hi, f_hi, dphi_hi, dphi0, t_best, f_best;
    int ls_phase, ls_iter, ls_done, ls_ok;
    int hist_n, hist_pos;
    int trainable[HMAX];
    int box[HMAX];
    int fail, done, status, n_eval, n_eval_opt, iter, phase, want_grad;
    int tile;
};

template <int D>
struct Ctx {
    float *xs, *xsc, *y, *z, *alpha, *Ad, *LinvT, *tmp;
    Shared* sh;
    float *U, *Dinv, *DinvT, *Vs;
    int N, NB, Npad, P;
    int tid, lane, w, h, g;
    float sf2, sn2;
    int kern;
};

// ---------------------------------------------------------------------------------------------
// K block (rows 32*bi.., cols 32*bj..) in acc layout, built from the scaled coordinates in LDS.
// Padding rows/cols (index >= N) form an identity block so that they add exactly 0 to the log-det,
// the quadratic form and the gradient ("padding identity").
// ---------------------------------------------------------------------------------------------
template <int D, int KERN>
__device__ __forceinline__ f32x16 kblock_t(const Ctx<D>& c, int bi, int bj) {
    const int q = 32 * bj + c.g;
    float xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = c.xsc[d * c.Npad + q];
    f32x16 out;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * bi + 8 * qq + 4 * c.h;
        float r2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            f32x4 xp = *reinterpret_cast<const f32x4*>(c.xsc + d * c.Npad + p0);
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

template <int D>
__device__ __forceinline__ f32x16 kblock(const Ctx<D>& c, int bi, int bj) {
    switch (c.kern) {
        case 0: return kblock_t<D, 0>(c, bi, bj);
        case 1: return kblock_t<D, 1>(c, bi, bj);
        case 2: return kblock_t<D, 2>(c, bi, bj);
        default: return kblock_t<D, 3>(c, bi, bj);
    }
}

// cross-covariance block: rows = observations 32*bj.., cols = prediction points 32*pc.. (xq scaled)
template <int D, int KERN>
__device__ __forceinline__ f32x16 ksblock_t(const Ctx<D>& c, int bj, const float (&xq)[D], bool qvalid) {
    f32x16 out;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * bj + 8 * qq + 4 * c.h;
        float r2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            f32x4 xp = *reinterpret_cast<const f32x4*>(c.xsc + d * c.Npad + p0);
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

template <int D>
__device__ __forceinline__ f32x16 ksblock(const Ctx<D>& c, int bj, const float (&xq)[D], bool qvalid) {
    switch (c.kern) {
        case 0: return ksblock_t<D, 0>(c, bj, xq, qvalid);
        case 1: return ksblock_t<D, 1>(c, bj, xq, qvalid);
        case 2: return ksblock_t<D, 2>(c, bj, xq, qvalid);
        default: return ksblock_t<D, 3>(c, bj, xq, qvalid);
    }
}

// ---------------------------------------------------------------------------------------------
// gradient contraction of one (K^-1)_ab block held in registers.
//   acc_l[d] += wgt * Q * g(r) * (scaled diff_d)^2 ; acc_sf += wgt * Q * kf ; acc_sn += Q on the diagonal
// ---------------------------------------------------------------------------------------------
template <int D, int KERN>
__device__ __forceinline__ void contract_t(const Ctx<D>& c, const f32x16& kinv, int ba, int bb, float wgt,
                                           float (&accl)[D], float& accsf, float& accsn) {
    const int q = 32 * bb + c.g;
    float xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = c.xsc[d * c.Npad + q];
    const float aq = c.alpha[q];
    const bool qv = q < c.N;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int p0 = 32 * ba + 8 * qq + 4 * c.h;
        float r2[4] = {0.f, 0.f, 0.f, 0.f};
        float d2[D][4];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            f32x4 xp = *reinterpret_cast<const f32x4*>(c.xsc + d * c.Npad + p0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float df = xp[e] - xq[d];
                d2[d][e] = df * df;
                r2[e] += d2[d][e];
            }
        }
        f32x4 ap = *reinterpret_cast<const f32x4*>(c.alpha + p0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = p0 + e;
            float kf, gg;
            kfun<KERN>(r2[e], kf, gg);
            float Q = kinv[4 * qq + e] - ap[e] * aq;
            Q = (qv && p < c.N) ? Q : 0.f;
            const float wq = wgt * Q;
            accsf = fmaf(wq, kf, accsf);
            const float wg = wq * gg;
#pragma unroll
            for (int d = 0; d < D; ++d) accl[d] = fmaf(wg, d2[d][e], accl[d]);
            if (p == q) accsn += Q;
        }
    }
}

template <int D>
__device__ __forceinline__ void contract(const Ctx<D>& c, const f32x16& kinv, int ba, int bb, float wgt,
                                         float (&accl)[D], float& accsf, float& accsn) {
    switch (c.kern) {
        case 0: contract_t<D, 0>(c, kinv, ba, bb, wgt, accl, accsf, accsn); break;
        case 1: contract_t<D, 1>(c, kinv, ba, bb, wgt, accl, accsf, accsn); break;
        case 2: contract_t<D, 2>(c, kinv, ba, bb, wgt, accl, accsf, accsn); break;
        default: contract_t<D, 3>(c, kinv, ba, bb, wgt, accl, accsf, accsn); break;
    }
}

// ---------------------------------------------------------------------------------------------
// 32x32 diagonal block: Cholesky W = L L^T and X = L^-1, one wave, registers + cross-lane reads.
// In : W (acc layout).  Out: S1 = X in acc layout, S2 = X^T in acc layout, Ad = X row-major [32][33]
//      logsum = sum log L_kk, bad = 1 when a pivot is not positive (or NaN).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void diag_factor(const f32x16& W, float* __restrict__ Ad, int lane, f32x16& S1, f32x16& S2,
                                         double& logsum, int& bad) {
    const int h = lane >> 5, g = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) Ad[rho(r, h) * 33 + g] = W[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float a[32];
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) a[cc] = Ad[g * 33 + cc];   // lane g (and its mirror g+32) holds row g
    int isbad = 0;
    float mydiag = 1.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        float dk = readlane_f(a[k], k);
        if (!(dk > 0.f)) { isbad = 1; dk = 1.f; }
        const float sd = sqrtf(dk);
        const float iv = 1.0f / sd;
        mydiag = (g == k) ? sd : mydiag;
        a[k] *= iv;                                  // column k of L (rows >= k)
#pragma unroll
        for (int cc = k + 1; cc < 32; ++cc) {
            const float lck = readlane_f(a[k], cc);  // L[cc][k]
            a[cc] = fmaf(-a[k], lck, a[cc]);
        }
    }
    // X = L^-1, lane g computes column g of X
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float s = (g == i) ? 1.f : 0.f;
#pragma unroll
        for (int cc = 0; cc < i; ++cc) s = fmaf(-readlane_f(a[cc], i), x[cc], s);   // L[i][cc]
        x[i] = s / readlane_f(a[i], i);
    }
    // S1[r] on lane (h,g) = X[rho(r,h)][g]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int r0 = (r & 3) + 8 * (r >> 2);
        S1[r] = h ? x[r0 + 4] : x[r0];
    }
    __builtin_amdgcn_wave_barrier();
    if (h == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) Ad[i * 33 + g] = x[i];          // row-major X
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 16; ++r) S2[r] = Ad[g * 33 + rho(r, h)];      // X^T in acc layout
    // sum log L_kk in fp64: lane k (< 32) owns pivot k
    double lg = (h == 0) ? log((double)mydiag) : 0.0;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) lg += __shfl_xor(lg, off);
    logsum = __shfl(lg, 0);
    bad = isbad;
}

// ---------------------------------------------------------------------------------------------
// phase 1: blocked Cholesky (upper, K = U^T U) with fused K build and forward solve z = L^-1 y
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void phase_potrf(Ctx<D>& c) {
    Shared* sh = c.sh;
    const int NB = c.NB, lane = c.lane, w = c.w;
    if (c.tid == 0) { sh->logdet = 0.0; sh->fail = 0; }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        const int nOff = NB - 1 - j;             // off-diagonal blocks of row j: i = j+1+idx
        // ---- step 1: wave 0 -> diagonal block; waves 1..3 -> first pair of off-diagonal blocks
        f32x16 W0 = zero16(), W1 = zero16();
        int i0 = -1, i1 = -1;
        if (w == 0) {
            float tp = 0.f;
            for (int k = 0; k < j; ++k) {
                f32x16 A = load_blk(c.U + (size_t)(k * NB + j) * BLK, lane);
                mma_blk(W0, A, A);
#pragma unroll
                for (int r = 0; r < 16; ++r) tp = fmaf(A[r], c.z[32 * k + rho(r, c.h)], tp);
            }
            f32x16 Kd = kblock<D>(c, j, j);
            f32x16 Wd = Kd - W0;
            f32x16 S1, S2;
            double ls; int bad;
            diag_factor(Wd, c.Ad, lane, S1, S2, ls, bad);
            store_blk(c.Dinv + (size_t)j * BLK, lane, S1);
            store_blk(c.DinvT + (size_t)j * BLK, lane, S2);
            store_blk(c.LinvT, lane, S2);
            // z_j = L_jj^-1 (y_j - t_j)
            const float t = xhalf_sum(tp);
            if (c.h == 0) c.tmp[c.g] = c.y[32 * j + c.g] - t;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float zz = 0.f;
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) zz = fmaf(c.Ad[c.g * 33 + cc], c.tmp[cc], zz);
            if (c.h == 0) c.z[32 * j + c.g] = zz;
            if (lane == 0) { sh->logdet += ls; if (bad) sh->fail = 1; }
            W0 = zero16(); W1 = zero16();
        } else {
            const int idx0 = 2 * (w - 1), idx1 = idx0 + 1;
            if (idx0 < nOff) i0 = j + 1 + idx0;
            if (idx1 < nOff) i1 = j + 1 + idx1;
            if (i0 >= 0) {
                for (int k = 0; k < j; ++k) {
                    f32x16 A = load_blk(c.U + (size_t)(k * NB + j) * BLK, lane);
                    f32x16 B0 = load_blk(c.U + (size_t)(k * NB + i0) * BLK, lane);
                    mma_blk(W0, A, B0);
                    if (i1 >= 0) {
                        f32x16 B1 = load_blk(c.U + (size_t)(k * NB + i1) * BLK, lane);
                        mma_blk(W1, A, B1);
                    }
                }
                W0 = kblock<D>(c, j, i0) - W0;
                if (i1 >= 0) W1 = kblock<D>(c, j, i1) - W1;
            }
        }
        __syncthreads();
        if (sh->fail) break;
        // ---- step 2: apply L_jj^-1 and store; then the remaining pairs over all 4 waves
        const f32x16 Lop = load_blk(c.LinvT, lane);
        if (i0 >= 0) {
            f32x16 Uo = zero16();
            mma_blk(Uo, Lop, W0);
            store_blk(c.U + (size_t)(j * NB + i0) * BLK, lane, Uo);
            if (i1 >= 0) {
                f32x16 Uo1 = zero16();
                mma_blk(Uo1, Lop, W1);
                store_blk(c.U + (size_t)(j * NB + i1) * BLK, lane, Uo1);
            }
        }
        for (int idx = 2 * (NW - 1) + 2 * w; idx < nOff; idx += 2 * NW) {
            const int a0 = j + 1 + idx;
            const int a1 = (idx + 1 < nOff) ? a0 + 1 : -1;
            f32x16 V0 = zero16(), V1 = zero16();
            for (int k = 0; k < j; ++k) {
                f32x16 A = load_blk(c.U + (size_t)(k * NB + j) * BLK, lane);
                f32x16 B0 = load_blk(c.U + (size_t)(k * NB + a0) * BLK, lane);
                mma_blk(V0, A, B0);
                if (a1 >= 0) {
                    f32x16 B1 = load_blk(c.U + (size_t)(k * NB + a1) * BLK, lane);
                    mma_blk(V1, A, B1);
                }
            }
            V0 = kblock<D>(c, j, a0) - V0;
            f32x16 Uo = zero16();
            mma_blk(Uo, Lop, V0);
            store_blk(c.U + (size_t)(j * NB + a0) * BLK, lane, Uo);
            if (a1 >= 0) {
                V1 = kblock<D>(c, j, a1) - V1;
                f32x16 Uo1 = zero16();
                mma_blk(Uo1, Lop, V1);
                store_blk(c.U + (size_t)(j * NB + a1) * BLK, lane, Uo1);
            }
        }
        __syncthreads();
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// phase 2: M = L^-1 by block columns (one wave per column, no inter-wave dependency) and
//          alpha = M^T z accumulated as each M_ij is produced.
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void phase_trtri(Ctx<D>& c) {
    const int NB = c.NB, lane = c.lane;
    for (int j0 = 0; j0 < NB; j0 += NW) {
        // snake assignment balances the triangular column costs over the 4 waves
        const int rnd = j0 / NW;
        const int j = j0 + ((rnd & 1) ? (NW - 1 - c.w) : c.w);
        if (j >= NB) continue;
        const f32x16 Mjj = load_blk(c.Dinv + (size_t)j * BLK, lane);
        float ap = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ap = fmaf(Mjj[r], c.z[32 * j + rho(r, c.h)], ap);
        for (int i = j + 1; i < NB; ++i) {
            f32x16 acc = zero16();
            {
                f32x16 A = load_blk(c.U + (size_t)(j * NB + i) * BLK, lane);
                mma_blk(acc, A, Mjj);
            }
            for (int k = j + 1; k < i; ++k) {
                f32x16 A = load_blk(c.U + (size_t)(k * NB + i) * BLK, lane);
                f32x16 B = load_blk(c.U + (size_t)(k * NB + j) * BLK, lane);   // M_kj (lower storage)
                mma_blk(acc, A, B);
            }
            const f32x16 Lop = load_blk(c.DinvT + (size_t)i * BLK, lane);
            f32x16 Mij = zero16();
            mma_blk(Mij, Lop, acc);
            Mij = -Mij;
            store_blk(c.U + (size_t)(i * NB + j) * BLK, lane, Mij);
#pragma unroll
            for (int r = 0; r < 16; ++r) ap = fmaf(Mij[r], c.z[32 * i + rho(r, c.h)], ap);
        }
        const float a = xhalf_sum(ap);
        if (c.h == 0) c.alpha[32 * j + c.g] = a;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// phase 3: (K^-1)_ab = sum_{c>=a} M_ca^T M_cb, contracted in registers with dK/dtheta.
// Writes sh->gth (dNLL/dtheta).
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void phase_grad(Ctx<D>& c) {
    Shared* sh = c.sh;
    const int NB = c.NB, lane = c.lane;
    float accl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) accl[d] = 0.f;
    float accsf = 0.f, accsn = 0.f;
    int pair = 0;
    for (int a = 0; a < NB; ++a) {
        for (int b = 0; b <= a; ++b, ++pair) {
            if ((pair & (NW - 1)) != c.w) continue;
            f32x16 acc = zero16();
            if (a == b) {
                f32x16 A = load_blk(c.Dinv + (size_t)a * BLK, lane);
                mma_blk(acc, A, A);
                for (int cc = a + 1; cc < NB; ++cc) {
                    f32x16 A2 = load_blk(c.U + (size_t)(cc * NB + a) * BLK, lane);
                    mma_blk(acc, A2, A2);
                }
            } else {
                {
                    f32x16 A = load_blk(c.Dinv + (size_t)a * BLK, lane);
                    f32x16 B = load_blk(c.U + (size_t)(a * NB + b) * BLK, lane);
                    mma_blk(acc, A, B);
                }
                for (int cc = a + 1; cc < NB; ++cc) {
                    f32x16 A = load_blk(c.U + (size_t)(cc * NB + a) * BLK, lane);
                    f32x16 B = load_blk(c.U + (size_t)(cc * NB + b) * BLK, lane);
                    mma_blk(acc, A, B);
                }
            }
            contract<D>(c, acc, a, b, (a == b) ? 1.f : 2.f, accl, accsf, accsn);
        }
    }
    // wave reduction (doubles), then across waves through LDS
    double v[D + 2];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = (double)accl[d];
    v[D] = (double)accsf;
    v[D + 1] = (double)accsn;
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
            if (i < D) {
                // scaled diff^2 already carries 1/l^2; dk/dl = g * diff^2 / l^3
                sh->gth[i] = 0.5 * (double)c.sf2 * s / sh->theta[i];
            } else if (i == D) {
                sh->gth[i] = 0.5 * s;            // dK/dsf2 = kf (without sf2)
            } else {
                sh->gth[i] = 0.5 * s;
            }
        }
    }
    __syncthreads();
}

// quadratic form + assemble NLL (all threads)
template <int D>
__device__ __forceinline__ void finish_nll(Ctx<D>& c) {
    Shared* sh = c.sh;
    double q = 0.0;
    for (int p = c.tid; p < c.N; p += NT) { const double zz = (double)c.z[p]; q += zz * zz; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    if (c.lane == 0) sh->red[c.w][7] = q;
    __syncthreads();
    if (c.tid == 0) {
        double s = 0.0;
        for (int ww = 0; ww < NW; ++ww) s += sh->red[ww][7];
        sh->nll = 0.5 * s + sh->logdet + 0.5 * (double)c.N * 1.8378770664093453;   // log(2 pi)
    }
    __syncthreads();
}

// one objective (+ gradient) evaluation at sh->theta.  On return sh->nll, sh->gth, sh->fail are set.
template <int D>
__device__ __forceinline__ void evaluate(Ctx<D>& c, bool want_grad) {
    Shared* sh = c.sh;
    __syncthreads();
    float invl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) invl[d] = (float)(1.0 / sh->theta[d]);
    c.sf2 = (float)sh->theta[D];
    c.sn2 = (float)sh->theta[D + 1];
    for (int idx = c.tid; idx < c.Npad; idx += NT) {
#pragma unroll
        for (int d = 0; d < D; ++d) c.xsc[d * c.Npad + idx] = c.xs[d * c.Npad + idx] * invl[d];
    }
    __syncthreads();
    phase_potrf<D>(c);
    if (sh->fail) {
        if (c.tid == 0) { sh->nll = __builtin_inf(); for (int i = 0; i < D + 2; ++i) sh->gth[i] = 0.0; }
        __syncthreads();
        return;
    }
    finish_nll<D>(c);
    if (want_grad) {
        phase_trtri<D>(c);
        phase_grad<D>(c);
    }
    if (c.tid == 0) {
        sh->n_eval += 1;
        if (!(sh->nll == sh->nll)) sh->fail = 1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// prediction: V = L^-1 K_* by 32-column chunks (one wave per chunk), f* = V^T z,
// f*_var = sf2 - colsum(V^2), y_var = f*_var + sn2   (GPSat/models/gpflow_models.py:229-243)
// Requires U, DinvT, z of a successful phase_potrf at the final parameters.
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void predict_tile(Ctx<D>& c, const float* __restrict__ Xs, float* __restrict__ fm,
                             float* __restrict__ fv, float* __restrict__ yv, const float (&invl)[D]) {
    const int NB = c.NB, lane = c.lane;
    const int PC = (c.P + 31) / 32;
    float* Vw = c.Vs + (size_t)c.w * NB * BLK;
    for (int pc = c.w; pc < PC; pc += NW) {
        const int q = 32 * pc + c.g;
        const bool qv = q < c.P;
        float xq[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xq[d] = qv ? Xs[(size_t)q * D + d] * invl[d] : 0.f;
        float vs = 0.f, ms = 0.f;
        for (int j = 0; j < NB; ++j) {
            f32x16 acc = zero16();
            for (int k = 0; k < j; ++k) {
                f32x16 A = load_blk(c.U + (size_t)(k * NB + j) * BLK, lane);
                f32x16 B = load_blk(Vw + (size_t)k * BLK, lane);
                mma_blk(acc, A, B);
            }
            f32x16 Wb = ksblock<D>(c, j, xq, qv) - acc;
            const f32x16 Lop = load_blk(c.DinvT + (size_t)j * BLK, lane);
            f32x16 V = zero16();
            mma_blk(V, Lop, Wb);
            store_blk(Vw + (size_t)j * BLK, lane, V);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vs = fmaf(V[r], V[r], vs);
                ms = fmaf(V[r], c.z[32 * j + rho(r, c.h)], ms);
            }
        }
        vs = xhalf_sum(vs);
        ms = xhalf_sum(ms);
        if (c.h == 0 && qv) {
            const float var = c.sf2 - vs;
            fm[q] = ms;
            fv[q] = var;
            yv[q] = var + c.sn2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// parameter transforms (reference GPSat/utils.py:2320-2400,
// GPSat/models/gpflow_models.py:490-494): box -> lo + (hi-lo) sigmoid(u); else softplus(u) + shift
// ---------------------------------------------------------------------------------------------
__device__ inline double softplus_d(double x) { return log1p(exp(-fabs(x))) + fmax(x, 0.0); }

__device__ inline double theta_of_u(const Shared* sh, int i, double u) {
    if (sh->box[i]) return sh->lo[i] + (sh->hi[i] - sh->lo[i]) / (1.0 + exp(-u));
    return softplus_d(u) + sh->shift[i];
}

__device__ inline double u_of_theta(const Shared* sh, int i, double th) {
    if (sh->box[i]) {
        const double lo = sh->lo[i], hi = sh->hi[i];
        double t = (th - lo) / (hi - lo);
        t = fmin(fmax(t, 1e-15), 1.0 - 1e-15);
        return log(t / (1.0 - t));
    }
    double y = th - sh->shift[i];
    if (y < 1e-300) y = 1e-300;
    if (y > 34.0) return y;
    if (y < 1e-15) return log(y);
    return log(-expm1(-y)) + y;
}

__device__ inline double dtheta_du(const Shared* sh, int i, double th) {
    if (sh->box[i]) return (th - sh->lo[i]) * (sh->hi[i] - th) / (sh->hi[i] - sh->lo[i]);
    return -expm1(-(th - sh->shift[i]));
}

// thread 0: trial u -> theta for the next evaluation
__device__ inline void set_trial(Shared* sh, int H, const double* u) {
    for (int i = 0; i < H; ++i) {
        sh->ut[i] = u[i];
        if (sh->trainable[i]) sh->theta[i] = theta_of_u(sh, i, u[i]);
    }
}

// thread 0: after an evaluation, chain the gradient to u-space at the trial point
__device__ inline void fetch_trial(Shared* sh, int H) {
    sh->ft = sh->fail ? __builtin_inf() : sh->nll;
    for (int i = 0; i < H; ++i)
        sh->gt[i] = (sh->trainable[i] && !sh->fail) ? sh->gth[i] * dtheta_du(sh, i, sh->theta[i]) : 0.0;
}

// L-BFGS two-loop recursion (thread 0): d = -H g
__device__ inline void lbfgs_direction(Shared* sh, int H) {
    double q[HMAX], al[MH];
    for (int i = 0; i < H; ++i) q[i] = sh->g[i];
    const int n = sh->hist_n;
    for (int m = 0; m < n; ++m) {
        const int idx = (sh->hist_pos - 1 - m + 2 * MH) % MH;
        double a = 0.0;
        for (int i = 0; i < H; ++i) a += sh->S[idx][i] * q[i];
        a *= sh->rho_[idx];
        al[m] = a;
        for (int i = 0; i < H; ++i) q[i] -= a * sh->Y[idx][i];
    }
    if (n > 0) {
        const int idx = (sh->hist_pos - 1 + MH) % MH;
        double sy = 0.0, yy = 0.0;
        for (int i = 0; i < H; ++i) { sy += sh->S[idx][i] * sh->Y[idx][i]; yy += sh->Y[idx][i] * sh->Y[idx][i]; }
        const double gam = sy / yy;
        for (int i = 0; i < H; ++i) q[i] *= gam;
    }
    for (int m = n - 1; m >= 0; --m) {
        const int idx = (sh->hist_pos - 1 - m + 2 * MH) % MH;
        double b = 0.0;
        for (int i = 0; i < H; ++i) b += sh->Y[idx][i] * q[i];
        b *= sh->rho_[idx];
        for (int i = 0; i < H; ++i) q[i] += (al[m] - b) * sh->S[idx][i];
    }
    for (int i = 0; i < H; ++i) sh->d[i] = -q[i];
}

__device__ inline double cubic_min(double a, double fa, double da, double b, double fb, double db) {
    // minimiser of the cubic interpolating (a,fa,da), (b,fb,db); falls back to bisection
    const double d1 = da + db - 3.0 * (fa - fb) / (a - b);
    const double rad = d1 * d1 - da * db;
    if (!(rad >= 0.0)) return 0.5 * (a + b);
    double d2 = sqrt(rad);
    if (b < a) d2 = -d2;
    const double den = db - da + 2.0 * d2;
    if (den == 0.0) return 0.5 * (a + b);
    const double t = b - (b - a) * ((db + d2 - d1) / den);
    if (!(t == t)) return 0.5 * (a + b);
    return t;
}

// strong-Wolfe line search step (thread 0).  Called after each trial evaluation.
// Sets sh->ls_done (1 accepted / 2 failed) or the next sh->t.
__device__ inline void ls_step(Shared* sh, int H, int max_ls) {
    const double c1 = 1e-4, c2 = 0.9;
    const double t = sh->t, ft = sh->ft;
    double dphit = 0.0;
    for (int i = 0; i < H; ++i) dphit += sh->gt[i] * sh->d[i];
    const bool finite = (ft == ft) && (ft < 1e300);
    const bool armijo = finite && (ft <= sh->f + c1 * t * sh->dphi0);
    if (armijo && ft < sh->f_best) { sh->f_best = ft; sh->t_best = t; }
    sh->ls_iter += 1;
    if (armijo && fabs(dphit) <= -c2 * sh->dphi0) { sh->ls_done = 1; return; }
    if (sh->ls_iter >= max_ls) { sh->ls_done = (armijo ? 1 : 2); return; }
    if (sh->ls_phase == 0) {
        if (!armijo || (sh->ls_iter > 1 && ft >= sh->f_prev)) {
            sh->t_lo = sh->t_prev; sh->f_lo = sh->f_prev; sh->dphi_lo = sh->dphi_prev;
            sh->t_hi = t; sh->f_hi = ft; sh->dphi_hi = dphit;
            sh->ls_phase = 1;
        } else if (dphit >= 0.0) {
            sh->t_lo = t; sh->f_lo = ft; sh->dphi_lo = dphit;
            sh->t_hi = sh->t_prev; sh->f_hi = sh->f_prev; sh->dphi_hi = sh->dphi_prev;
            sh->ls_phase = 1;
        } else {
            sh->t_prev = t; sh->f_prev = ft; sh->dphi_prev = dphit;
            sh->t = 2.0 * t;
            return;
        }
    } else {
        if (!armijo || ft >= sh->f_lo) {
            sh->t_hi = t; sh->f_hi = ft; sh->dphi_hi = dphit;
        } else {
            if (dphit * (sh->t_hi - sh->t_lo) >= 0.0) { sh->t_hi = sh->t_lo; sh->f_hi = sh->f_lo; sh->dphi_hi = sh->dphi_lo; }
            sh->t_lo = t; sh->f_lo = ft; sh->dphi_lo = dphit;
        }
    }
    // next trial inside (lo, hi)
    const double lo = sh->t_lo, hi = sh->t_hi;
    double tn;
    const bool hi_finite = (sh->f_hi == sh->f_hi) && (sh->f_hi < 1e300);
    if (hi_finite) tn = cubic_min(lo, sh->f_lo, sh->dphi_lo, hi, sh->f_hi, sh->dphi_hi);
    else tn = 0.5 * (lo + hi);
    const double a = fmin(lo, hi), b = fmax(lo, hi), wdt = b - a;
    if (!(tn > a + 0.1 * wdt && tn < b - 0.1 * wdt)) tn = 0.5 * (a + b);
    if (wdt < 1e-12 * fmax(1.0, b)) { sh->ls_done = (armijo ? 1 : 2); return; }
    sh->t = tn;
}

// ---------------------------------------------------------------------------------------------
// optimiser driver (thread 0): a state machine advanced once per objective evaluation, so that the
// kernel has ONE inlined call site of evaluate().
// ---------------------------------------------------------------------------------------------
enum { PH_INIT = 0, PH_LS = 1, PH_ADAM = 2, PH_FINAL = 3, PH_EXIT = 4 };

struct OptCfg { int optimiser, max_iter, max_ls, want_grad_out; double ftol, gtol, adam_lr; };

// the accepted point is sh->u; decide whether the factorisation in memory already belongs to it
__device__ inline void opt_finish(Shared* sh, int H, const OptCfg& o, bool factor_is_current) {
    sh->n_eval_opt = sh->n_eval;
    if (factor_is_current && !sh->fail) { sh->phase = PH_EXIT; return; }
    set_trial(sh, H, sh->u);
    sh->want_grad = o.want_grad_out;
    sh->phase = PH_FINAL;
}

__device__ inline void opt_start_iteration(Shared* sh, int H, const OptCfg& o) {
    if (o.optimiser == 2) {
        const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
        const int k = sh->iter + 1;
        double un[HMAX];
        for (int i = 0; i < H; ++i) {
            sh->m1[i] = b1 * sh->m1[i] + (1 - b1) * sh->g[i];
            sh->m2[i] = b2 * sh->m2[i] + (1 - b2) * sh->g[i] * sh->g[i];
            const double mh = sh->m1[i] / (1 - pow(b1, (double)k)), vh = sh->m2[i] / (1 - pow(b2, (double)k));
            un[i] = sh->u[i] - (sh->trainable[i] ? o.adam_lr * mh / (sqrt(vh) + eps) : 0.0);
        }
        set_trial(sh, H, un);
        sh->phase = PH_ADAM;
        return;
    }
    lbfgs_direction(sh, H);
    double dphi0 = 0.0, gn = 0.0;
    for (int i = 0; i < H; ++i) { dphi0 += sh->g[i] * sh->d[i]; gn += sh->g[i] * sh->g[i]; }
    if (!(dphi0 < 0.0)) {      // not a descent direction: restart from steepest descent
        sh->hist_n = 0;
        for (int i = 0; i < H; ++i) sh->d[i] = -sh->g[i];
        dphi0 = -gn;
    }
    if (gn == 0.0) { sh->status = 0; opt_finish(sh, H, o, true); return; }
    sh->dphi0 = dphi0;
    sh->t = (sh->hist_n == 0) ? fmin(1.0, 1.0 / sqrt(gn)) : 1.0;
    sh->ls_phase = 0; sh->ls_iter = 0; sh->ls_done = 0;
    sh->t_prev = 0.0; sh->f_prev = sh->f; sh->dphi_prev = dphi0;
    sh->t_best = 0.0; sh->f_best = sh->f;
    double un[HMAX];
    for (int i = 0; i < H; ++i) un[i] = sh->u[i] + sh->t * sh->d[i];
    set_trial(sh, H, un);
    sh->phase = PH_LS;
}

__device__ inline void opt_advance(Shared* sh, int H, const OptCfg& o) {
    switch (sh->phase) {
        case PH_INIT: {
            set_trial(sh, H, sh->u);
            fetch_trial(sh, H);
            sh->f = sh->ft;
            for (int i = 0; i < H; ++i) sh->g[i] = sh->gt[i];
            if (sh->fail) { sh->status = 2; sh->n_eval_opt = sh->n_eval; sh->phase = PH_EXIT; return; }
            sh->status = 1;
            opt_start_iteration(sh, H, o);
            return;
        }
        case PH_ADAM: {
            fetch_trial(sh, H);
            if (sh->fail) { sh->status = 2; opt_finish(sh, H, o, false); return; }
            sh->f = sh->ft;
            for (int i = 0; i < H; ++i) { sh->u[i] = sh->ut[i]; sh->g[i] = sh->gt[i]; }
            sh->iter += 1;
            if (sh->iter >= o.max_iter) { sh->status = 1; opt_finish(sh, H, o, true); return; }
            opt_start_iteration(sh, H, o);
            return;
        }
        case PH_LS: {
            fetch_trial(sh, H);
            ls_step(sh, H, o.max_ls);
            if (!sh->ls_done) {
                double un[HMAX];
                for (int i = 0; i < H; ++i) un[i] = sh->u[i] + sh->t * sh->d[i];
                set_trial(sh, H, un);
                return;
            }
            if (sh->ls_done == 1) {
                // accept the trial point (the last evaluated one)
                double sy = 0.0, yy = 0.0, gmax = 0.0;
                double sv[HMAX], yvv[HMAX];
                for (int i = 0; i < H; ++i) {
                    sv[i] = sh->ut[i] - sh->u[i];
                    yvv[i] = sh->gt[i] - sh->g[i];
                    sy += sv[i] * yvv[i];
                    yy += yvv[i] * yvv[i];
                }
                if (sy > 1e-10 * yy && yy > 0.0) {
                    const int pos = sh->hist_pos;
                    for (int i = 0; i < H; ++i) { sh->S[pos][i] = sv[i]; sh->Y[pos][i] = yvv[i]; }
                    sh->rho_[pos] = 1.0 / sy;
                    sh->hist_pos = (pos + 1) % MH;
                    if (sh->hist_n < MH) sh->hist_n += 1;
                }
                const double fold = sh->f, fnew = sh->ft;
                sh->f = fnew;
                for (int i = 0; i < H; ++i) { sh->u[i] = sh->ut[i]; sh->g[i] = sh->gt[i]; gmax = fmax(gmax, fabs(sh->gt[i])); }
                sh->iter += 1;
                const double den = fmax(fmax(fabs(fold), fabs(fnew)), 1.0);
                if ((fold - fnew) <= o.ftol * den || gmax <= o.gtol) { sh->status = 0; opt_finish(sh, H, o, true); return; }
                if (sh->iter >= o.max_iter) { sh->status = 1; opt_finish(sh, H, o, true); return; }
                opt_start_iteration(sh, H, o);
                return;
            }
            // line search failed: no further decrease is resolvable at this precision
            if (sh->hist_n > 0 && sh->iter + 1 < o.max_iter) {
                sh->hist_n = 0;           // one restart with steepest descent from the accepted point
                sh->iter += 1;
                opt_start_iteration(sh, H, o);
                return;
            }
            sh->status = (sh->iter + 1 >= o.max_iter && sh->hist_n > 0) ? 1 : 0;
            opt_finish(sh, H, o, false);
            return;
        }
        default:  // PH_FINAL
            sh->phase = PH_EXIT;
            return;
    }
}

// ---------------------------------------------------------------------------------------------
// the persistent kernel
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(NT, 2) gp_tile_kernel(const KernelArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int H = D + 2;
    Ctx<D> c;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.w = c.tid >> 6;
    c.h = c.lane >> 5;
    c.g = c.lane & 31;
    c.kern = A.kernel;
    const int NPmax = A.NBmax * 32;
    Shared* sh = reinterpret_cast<Shared*>(smem);
    c.sh = sh;
    float* fp = reinterpret_cast<float*>(smem + ((sizeof(Shared) + 15) & ~size_t(15)));
    c.xs = fp; fp += D * NPmax;
    c.xsc = fp; fp += D * NPmax;
    c.y = fp; fp += NPmax;
    c.z = fp; fp += NPmax;
    c.alpha = fp; fp += NPmax;
    c.Ad = fp; fp += 32 * 33 + 3;   // 1059 floats; re-align below
    fp = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(fp) + 15) & ~uintptr_t(15));
    c.LinvT = fp; fp += BLK;
    c.tmp = fp; fp += 32;
    float* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
    OptCfg o;
    o.optimiser = A.optimiser; o.max_iter = A.max_iter; o.max_ls = A.max_ls; o.want_grad_out = A.grad != nullptr;
    o.ftol = A.ftol; o.gtol = A.gtol; o.adam_lr = A.adam_lr;

    for (;;) {
        __syncthreads();
        if (c.tid == 0) sh->tile = atomicAdd(A.queue, 1);
        __syncthreads();
        const int slot = sh->tile;
        if (slot >= A.T) break;
        const int t = A.order[slot];
        const long long o0 = A.obs_off[t], o1 = A.obs_off[t + 1];
        const long long p0 = A.pred_off[t], p1 = A.pred_off[t + 1];
        c.N = (int)(o1 - o0);
        c.P = (int)(p1 - p0);
        c.NB = (c.N + 31) / 32;
        c.Npad = c.NB * 32;
        const int NB = c.NB;
        c.U = ws;
        c.Dinv = ws + (size_t)NB * NB * BLK;
        c.DinvT = c.Dinv + (size_t)NB * BLK;
        c.Vs = c.DinvT + (size_t)NB * BLK;
        if (c.N == 0) {
            if (c.tid == 0) {
                A.status[t] = 4; A.n_eval[t] = 0; A.nll[t] = 0.0;
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
            continue;
        }
        // ---- stage tile data into LDS (SoA coordinates), zero padding
        for (int idx = c.tid; idx < c.Npad; idx += NT) {
            const bool v = idx < c.N;
#pragma unroll
            for (int d = 0; d < D; ++d) c.xs[d * c.Npad + idx] = v ? A.X[(size_t)(o0 + idx) * D + d] : 0.f;
            c.y[idx] = v ? A.y[o0 + idx] : 0.f;
            c.z[idx] = 0.f;
            c.alpha[idx] = 0.f;
        }
        if (c.tid == 0) {
            sh->n_eval = 0; sh->n_eval_opt = 0; sh->status = 5; sh->iter = 0; sh->hist_n = 0; sh->hist_pos = 0;
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
        __syncthreads();

        // ================= evaluate / advance loop (one inlined evaluate call site) =================
        for (;;) {
            evaluate<D>(c, sh->want_grad != 0);
            if (c.tid == 0) opt_advance(sh, H, o);
            __syncthreads();
            if (sh->phase == PH_EXIT) break;
        }

        // ================= outputs + prediction from the factorisation at the accepted parameters
        if (c.tid == 0) {
            int st = sh->status;
            if (sh->fail) st = (sh->nll == sh->nll) ? 2 : 3;
            A.status[t] = st;
            A.n_eval[t] = sh->n_eval_opt;
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
                for (int d = 0; d < D; ++d) invl[d] = (float)(1.0 / sh->theta[d]);
                predict_tile<D>(c, A.Xs + (size_t)p0 * D, A.f_mean + p0, A.f_var + p0, A.y_var + p0, invl);
            } else {
                for (long long q = p0 + c.tid; q < p1; q += NT) {
                    A.f_mean[q] = __builtin_nanf(""); A.f_var[q] = __builtin_nanf(""); A.y_var[q] = __builtin_nanf("");
                }
            }
        }
    }
}

size_t shared_bytes(int D, int NBmax) {
    size_t s = (sizeof(Shared) + 15) & ~size_t(15);
    const size_t NP = (size_t)NBmax * 32;
    s += sizeof(float) * (2 * D * NP + 3 * NP + 32 * 33 + 3 + 4 + BLK + 32);
    return (s + 15) & ~size_t(15);
}

size_t workspace_floats_per_wg(int NBmax) {
    // U/M square + Dinv + DinvT + per-wave V scratch
    return (size_t)BLK * ((size_t)NBmax * NBmax + 2 * (size_t)NBmax + (size_t)NW * NBmax);
}

hipError_t launch_tiles(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream) {
    switch (D) {
        case 1:
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gp_tile_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(gp_tile_kernel<1>, dim3(grid), dim3(NT), smem, stream, a);
            break;
        case 2:
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gp_tile_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(gp_tile_kernel<2>, dim3(grid), dim3(NT), smem, stream, a);
            break;
        case 3:
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gp_tile_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(gp_tile_kernel<3>, dim3(grid), dim3(NT), smem, stream, a);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gpsat
