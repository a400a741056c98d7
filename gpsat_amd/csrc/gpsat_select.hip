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
// Sub-chunk skipping.  A first kernel reduces every SEL_SUB consecutive rows to their per-column [min, max] box.  Before
// a wave streams a sub-chunk it tests, for its EB experts, whether ANY row of the box can satisfy ALL criteria; if none
// can for any of them, the sub-chunk is skipped (only its box, 1/64 of its bytes, was read).  The test is exact, not
// heuristic: IEEE subtraction, multiplication and addition are monotone, so the squared distance of the box's nearest
// point, accumulated in the reference's order, is a lower bound of every row's; 1-D comparisons are tested against the
// box ends.  Source tables that are ordered in time or along the satellite track (GPSat's are: daily files appended)
// skip almost everything outside an expert's window; on randomly ordered rows nothing is skipped and the box read
// costs 1.6 % more traffic.  Membership and output order are unchanged either way.
//
// Spatial binning.  Tables that are NOT ordered (and any table against 100 000 experts) are first sorted on the device by
// the grid cell of the criteria's columns (select_bin_rows: cell key per row, rocPRIM radix sort, gather into a permuted
// copy), so that every box of SEL_SUB consecutive rows is a small region and the skipping above works whatever order the
// rows came in; experts are dealt to the waves in the order of their own cells, so that the 8 experts of a wave want the
// same boxes.  The predicates run on the same fp64 values (membership unchanged bit for bit); the selected positions are
// mapped back to source rows and every expert's list is sorted ascending (select_unbin: rocPRIM segmented radix sort) --
// the reference's source row order (dataloader.py:2447).
//
// This is HBM/L2-bound streaming + integer compaction: no MFMA.  One wave owns EB experts and streams the point
// columns (SoA, coalesced 512-B wave reads, L2/MALL-resident across experts); matches are compacted with
// ballot / popcount / mbcnt, so no LDS and no barriers.  The rows are cut into chunks (grid.y) so that the launch
// has >> 256 workgroups; two passes: count per (expert, chunk), host scan, fill.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include "gpsat_kernels.h"

namespace gpsat {

constexpr int SEL_EB = 8;        // experts per wave
constexpr int SEL_NT = 256;      // threads per workgroup (4 independent waves)
constexpr int SEL_SUB = 1024;    // rows per bounding box (a multiple of 64; chunk_rows is a multiple of it)

// per-column [min, max] of every SEL_SUB consecutive rows: box[(sub * C + c) * 2 + {0, 1}]; one wave per sub-chunk
__global__ void __launch_bounds__(256) select_box_kernel(long long M, int C, const double* __restrict__ pts, double* __restrict__ box) {
    const int lane = threadIdx.x & 63;
    const long long sub = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long r0 = sub * SEL_SUB;
    if (r0 >= M) return;
    const long long r1 = min(M, r0 + SEL_SUB);
    for (int c = 0; c < C; ++c) {
        double mn = __builtin_inf(), mx = -__builtin_inf();
        for (long long i = r0 + lane; i < r1; i += 64) {
            const double x = pts[(size_t)c * M + i];
            mn = fmin(mn, x);            // fmin / fmax ignore a NaN operand; an all-NaN column leaves (+inf, -inf): never skipped
            mx = fmax(mx, x);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
        if (!(mn <= mx)) { mn = -__builtin_inf(); mx = __builtin_inf(); }       // no usable bound: the box excludes nothing
        if (lane == 0) { box[((size_t)sub * C + c) * 2] = mn; box[((size_t)sub * C + c) * 2 + 1] = mx; }
    }
}

// can a row with x in [mn, mx] satisfy  x <comp> y ?
__device__ __forceinline__ bool cmp1d_possible(int comp, double mn, double mx, double y) {
    switch (comp) {
        case 0: return mx >= y;
        case 1: return mx > y;
        case 2: return mn <= y && y <= mx;
        case 3: return mn < y;
        default: return mn <= y;
    }
}

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
    const int e0 = wave * SEL_EB;          // position in the dealing order (a.eorder), not an expert id
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
    int eid[SEL_EB];                       // the experts of this wave
#pragma unroll
    for (int e = 0; e < SEL_EB; ++e) eid[e] = a.eorder ? a.eorder[e0 + min(e, ne - 1)] : e0 + min(e, ne - 1);
#pragma unroll
    for (int e = 0; e < SEL_EB; ++e) {
        const int ee = eid[e];
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
    for (int e = 0; e < SEL_EB; ++e) cnt[e] = FILL ? a.off[(size_t)eid[e] * a.n_chunks + ch] : 0;
    for (long long sbeg = r_beg; sbeg < r_end; sbeg += SEL_SUB) {
      // ---- can any of this wave's experts select a row of this sub-chunk?  (wave-uniform arithmetic on its box)
      if (a.box) {
        const double* bx = a.box + (size_t)(sbeg / SEL_SUB) * a.C * 2;
        bool any = false;
#pragma unroll
        for (int e = 0; e < SEL_EB; ++e) {
            bool poss = e < ne;
#pragma unroll
            for (int k = 0; k < GPSAT_SEL_MAXCRIT; ++k) {
                if (k < a.n_crit) {
                    if (a.kind[k] == 0) {
                        const int cl = a.cols[k][0];
                        poss = poss && cmp1d_possible(a.comp[k], bx[cl * 2], bx[cl * 2 + 1], rhs[e][k]);
                    } else {
                        double sl = 0.0;     // squared distance of the box's nearest point, the reference's operation order
                        for (int m = 0; m < a.ncols[k]; ++m) {
                            const int cl = a.cols[k][m];
                            const double c0 = rc[e][k][m];
                            const double lo = __dsub_rn(bx[cl * 2], c0), hi = __dsub_rn(bx[cl * 2 + 1], c0);   // monotone in x
                            const double d = (lo > 0.0) ? lo : ((hi < 0.0) ? hi : 0.0);
                            sl = __dadd_rn(sl, __dmul_rn(d, d));
                        }
                        poss = poss && (a.comp[k] == 3 ? (sl < rhs[e][k]) : (sl <= rhs[e][k]));
                    }
                }
            }
            any = any || poss;
        }
        if (!any) continue;
      }
      const long long send = min(r_end, sbeg + SEL_SUB);
      for (long long base = sbeg; base < send; base += 64) {
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
    }
    if (!FILL && lane == 0) {
#pragma unroll
        for (int e = 0; e < SEL_EB; ++e)
            if (e < ne) a.counts[(size_t)eid[e] * a.n_chunks + ch] = cnt[e];
    }
}

hipError_t launch_select_boxes(long long M, int C, const double* pts, double* box, hipStream_t stream) {
    const long long nsub = (M + SEL_SUB - 1) / SEL_SUB;
    if (nsub > 0) hipLaunchKernelGGL(select_box_kernel, dim3((unsigned)((nsub + 3) / 4)), dim3(256), 0, stream, M, C, pts, box);
    return hipGetLastError();
}

int select_sub_rows() { return SEL_SUB; }

// ---- spatial binning -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) select_key_kernel(long long M, const double* __restrict__ pts, BinSpec b, unsigned* __restrict__ keys,
                                                         int* __restrict__ rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    unsigned key = 0;
    for (int d = 0; d < b.ndim; ++d) {
        const double x = pts[(size_t)b.col[d] * M + i];
        double cf = (x - b.origin[d]) * b.inv_cell[d];
        int cell = (cf >= 0.0) ? (int)fmin(cf, (double)(b.ncell[d] - 1)) : 0;       // NaN -> 0 (a NaN row matches nothing)
        key = key * (unsigned)b.ncell[d] + (unsigned)cell;
    }
    keys[i] = key;
    rows[i] = (int)i;
}

__global__ void __launch_bounds__(256) select_gather_kernel(long long M, int C, const double* __restrict__ pts, const int* __restrict__ perm,
                                                            double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int src = perm[i];
    for (int c = 0; c < C; ++c) out[(size_t)c * M + i] = pts[(size_t)c * M + src];
}

__global__ void __launch_bounds__(256) select_map_kernel(long long n, const int* __restrict__ perm, int* __restrict__ idx) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j < n) idx[j] = perm[idx[j]];
}

// temp == nullptr: only the size of the temporary storage is returned in temp_bytes
hipError_t select_bin_rows(long long M, int C, const double* pts, const BinSpec& b, unsigned* keys, unsigned* keys_out, int* rows,
                           int* perm, double* pts_perm, void* temp, size_t& temp_bytes, hipStream_t stream) {
    unsigned long long cells = 1;
    for (int d = 0; d < b.ndim; ++d) cells *= (unsigned long long)b.ncell[d];
    int bits = 1;
    while ((1ull << bits) < cells) ++bits;
    if (!temp) return rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys_out, rows, perm, (size_t)M, 0, bits, stream);
    const unsigned grid = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(select_key_kernel, dim3(grid), dim3(256), 0, stream, M, pts, b, keys, rows);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, rows, perm, (size_t)M, 0, bits, stream);   // stable
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(select_gather_kernel, dim3(grid), dim3(256), 0, stream, M, C, pts, perm, pts_perm);
    return hipGetLastError();
}

hipError_t select_unbin(int T, long long total, const unsigned* seg_off, const int* perm, int* idx, int* idx_out, void* temp,
                        size_t& temp_bytes, hipStream_t stream) {
    if (!temp) return rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, idx, idx_out, (unsigned)total, (unsigned)T, seg_off,
                                                         seg_off + 1, 0, 32, stream);
    hipLaunchKernelGGL(select_map_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, total, perm, idx);
    return rocprim::segmented_radix_sort_keys(temp, temp_bytes, idx, idx_out, (unsigned)total, (unsigned)T, seg_off, seg_off + 1, 0, 32,
                                              stream);
}

hipError_t launch_select(const SelectArgs& a, bool fill, hipStream_t stream) {
    const int waves = (a.T + SEL_EB - 1) / SEL_EB;
    const int grid = (waves + (SEL_NT / 64) - 1) / (SEL_NT / 64);
    if (fill) hipLaunchKernelGGL(select_kernel<true>, dim3(grid, a.n_chunks), dim3(SEL_NT), 0, stream, a);
    else hipLaunchKernelGGL(select_kernel<false>, dim3(grid, a.n_chunks), dim3(SEL_NT), 0, stream, a);
    return hipGetLastError();
}

}  // namespace gpsat
