// gpsat_select.hip -- batched tile selection on gfx950: which rows of a point table belong to each expert tile.
//
// Replaces, for all T experts at once, DataLoader.local_data_select (GPSat/dataloader.py:2352-2447) and the
// max_dist filter of PredictionLocations (GPSat/prediction_locations.py:18-43).  The predicates are evaluated in
// fp64 with exactly the reference's arithmetic, so membership is bit-exact:
//   * 1-D criterion :  x[col] <comp> (ref[col] + val)         (dataloader.py:2417-2421)
//   * ball criterion:  sum_k (x[c_k] - ref[c_k])^2  <= r*r     (KDTree.query_ball_point compares the squared distance,
//                      accumulated left to right from 0.0, with r*r; inclusive whatever `comp` says, :2439-2444)
//                      or  < r*r for prediction locations (strict, prediction_locations.py:37,43)
//   products and sums are NOT contracted into FMAs (__dmul_rn / __dadd_rn), as in the reference's host code.
// Output per expert: the selected row indices in SOURCE ROW ORDER (dataloader.py:2447), CSR-packed.
//
// This is HBM/L2-bound streaming + integer compaction: no MFMA.  One wave owns EB experts and streams the point
// columns (SoA, coalesced 512-B wave reads, L2/MALL-resident across experts); matches are compacted with
// ballot / popcount / mbcnt, so no LDS and no barriers.  The rows are cut into chunks (grid.y) so that the launch
// has >> 256 workgroups; two passes: count per (expert, chunk), host scan, fill.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gpsat_kernels.h"

namespace gpsat {

constexpr int SEL_EB = 8;        // experts per wave
constexpr int SEL_NT = 256;      // threads per workgroup (4 independent waves)

__device__ __forceinline__ bool cmp1d(int comp, double x, double y) {
    switch (comp) {
        case 0: return x >= y;
        case 1: return x > y;
        case 2: return x == y;
        case 3: return x < y;
        default: return x <= y;
    }
}

template <bool FILL>
__global__ void __launch_bounds__(SEL_NT) select_kernel(const SelectArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (SEL_NT / 64) + (threadIdx.x >> 6);
    const int e0 = wave * SEL_EB;
    if (e0 >= a.T) return;
    // blockIdx.y = row chunk: rows [r_beg, r_end); counts / offsets are kept per (expert, chunk) so that the fill pass
    // of every chunk knows where its rows go and the output stays in source row order
    const int ch = blockIdx.y;
    const long long r_beg = (long long)ch * a.chunk_rows;
    const long long r_end = min(a.M, r_beg + a.chunk_rows);
    const int ne = min(SEL_EB, a.T - e0);
    // per expert, per criterion: the right-hand side (1-D: ref + val; ball: r*r) -- wave-uniform
    double rhs[SEL_EB][GPSAT_SEL_MAXCRIT];
    double rc[SEL_EB][GPSAT_SEL_MAXCRIT][3];
#pragma unroll
    for (int e = 0; e < SEL_EB; ++e) {
        const int ee = e0 + min(e, ne - 1);
#pragma unroll
        for (int k = 0; k < GPSAT_SEL_MAXCRIT; ++k) {
            rhs[e][k] = 0.0;
            rc[e][k][0] = rc[e][k][1] = rc[e][k][2] = 0.0;
            if (k < a.n_crit) {
                if (a.kind[k] == 0) {
                    rhs[e][k] = __dadd_rn(a.refs[(size_t)ee * a.C + a.cols[k][0]], a.val[k]);
                } else {
                    rhs[e][k] = __dmul_rn(a.val[k], a.val[k]);
                    for (int m = 0; m < a.ncols[k]; ++m) rc[e][k][m] = a.refs[(size_t)ee * a.C + a.cols[k][m]];
                }
            }
        }
    }
    long long cnt[SEL_EB];
#pragma unroll
    for (int e = 0; e < SEL_EB; ++e) cnt[e] = FILL ? a.off[(size_t)(e0 + min(e, ne - 1)) * a.n_chunks + ch] : 0;
    for (long long base = r_beg; base < r_end; base += 64) {
        const long long i = base + lane;
        const bool inb = i < r_end;
        // this lane's point: the columns any criterion needs (at most MAXCRIT * 3 loads, L1/L2 hits)
        bool match[SEL_EB];
#pragma unroll
        for (int e = 0; e < SEL_EB; ++e) match[e] = inb && (e < ne);
#pragma unroll
        for (int k = 0; k < GPSAT_SEL_MAXCRIT; ++k) {
            if (k < a.n_crit) {
                if (a.kind[k] == 0) {
                    const double x = inb ? a.pts[(size_t)a.cols[k][0] * a.M + i] : 0.0;
#pragma unroll
                    for (int e = 0; e < SEL_EB; ++e) match[e] = match[e] && cmp1d(a.comp[k], x, rhs[e][k]);
                } else {
                    double x[3] = {0.0, 0.0, 0.0};
                    for (int m = 0; m < a.ncols[k]; ++m) x[m] = inb ? a.pts[(size_t)a.cols[k][m] * a.M + i] : 0.0;
#pragma unroll
                    for (int e = 0; e < SEL_EB; ++e) {
                        double s = 0.0;
                        for (int m = 0; m < a.ncols[k]; ++m) {
                            const double d = __dsub_rn(x[m], rc[e][k][m]);
                            s = __dadd_rn(s, __dmul_rn(d, d));
                        }
                        match[e] = match[e] && (a.comp[k] == 3 ? (s < rhs[e][k]) : (s <= rhs[e][k]));
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < SEL_EB; ++e) {
            const unsigned long long bal = __ballot(match[e]);
            if (FILL) {
                if (match[e]) {
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                    a.idx[cnt[e] + pos] = (int)i;
                }
            }
            cnt[e] += __popcll(bal);
        }
    }
    if (!FILL && lane == 0) {
#pragma unroll
        for (int e = 0; e < SEL_EB; ++e)
            if (e < ne) a.counts[(size_t)(e0 + e) * a.n_chunks + ch] = cnt[e];
    }
}

hipError_t launch_select(const SelectArgs& a, bool fill, hipStream_t stream) {
    const int waves = (a.T + SEL_EB - 1) / SEL_EB;
    const int grid = (waves + (SEL_NT / 64) - 1) / (SEL_NT / 64);
    if (fill) hipLaunchKernelGGL(select_kernel<true>, dim3(grid, a.n_chunks), dim3(SEL_NT), 0, stream, a);
    else hipLaunchKernelGGL(select_kernel<false>, dim3(grid, a.n_chunks), dim3(SEL_NT), 0, stream, a);
    return hipGetLastError();
}

}  // namespace gpsat
