// gpsat_post.hip -- post-processing rows of SURVEY.md section 8f on gfx950 (fp64, HBM / ALU-bound, no MFMA):
//   smooth_kernel : Gaussian-weighted smoothing of a hyper-parameter field over the expert locations
//                   (GPSat/postprocessing.py:22-52 gaussian_2d_weight): out_i = sum_j w_ij v_j / sum_j w_ij over the
//                   non-NaN v_j, w_ij = exp(-(((x_j-x_i)/lx)^2 + ((y_j-y_i)/ly)^2)/2), NaN when the weights sum to 0;
//   glue_kernel   : Gaussian-weighted averaging of overlapping local predictions per prediction location
//                   (GPSat/postprocessing.py:447-577): rows are pre-sorted by prediction location (CSR segments),
//                   w = prod_d normpdf(pred_d; xprt_d, sigma[row]), out[var][g] = sum w v / sum w.
// One wave per output element, lanes stride over the inputs (coalesced), fixed-order wave reduction => results are
// reproducible bit for bit (the reference accumulates sequentially; agreement is to fp64 rounding, ~1e-15 relative).
// smooth_kernel is bound by fp64 ALU work (two divisions and an exp per pair: 480 G pairs/s), not by its reads (24 B per pair
// out of L1 / L2): the LDS-tiled form (one thread per output, inputs staged in tiles of 256 and read by broadcast;
// scripts/experiments/r4_smooth_kernel_lds_tiled.patch) measured 390 G pairs/s at 131 072 locations and 17 instead of 229 at
// 4 096 (a quarter of the waves in flight, one dependent accumulation chain per thread) -- not adopted.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gpsat_kernels.h"

namespace gpsat {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256) smooth_kernel(int T, const double* __restrict__ x, const double* __restrict__ y,
                                                     const double* __restrict__ vals, double lx, double ly,
                                                     double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= T) return;
    const double x0 = x[i], y0 = y[i];
    double wv = 0.0, ws = 0.0;
    for (int j = lane; j < T; j += 64) {
        const double v = vals[j];
        if (v == v) {
            const double dx = (x[j] - x0) / lx, dy = (y[j] - y0) / ly;
            const double w = exp(-(dx * dx + dy * dy) / 2);
            wv += w * v;
            ws += w;
        }
    }
    wv = wave_sum(wv);
    ws = wave_sum(ws);
    if (lane == 0) out[i] = (ws == 0.0) ? __builtin_nan("") : wv / ws;
}

__global__ void __launch_bounds__(256) glue_kernel(int G, int ndim, int nvars, long long R, const long long* __restrict__ seg,
                                                   const double* __restrict__ pred, const double* __restrict__ xprt,
                                                   const double* __restrict__ vals, double sigma, const double* __restrict__ sigma_rows,
                                                   double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const long long a = seg[g], b = seg[g + 1];
    double ws = 0.0;
    double acc[GPSAT_GLUE_MAXVARS];
#pragma unroll
    for (int v = 0; v < GPSAT_GLUE_MAXVARS; ++v) acc[v] = 0.0;
    for (long long r = a + lane; r < b; r += 64) {
        const double sg = sigma_rows ? sigma_rows[r] : sigma;
        const double cnorm = 1.0 / (sg * 2.5066282746310002);       // 1 / (sigma sqrt(2 pi))
        double w = 1.0;
        for (int d = 0; d < ndim; ++d) {
            const double zz = (pred[(size_t)d * R + r] - xprt[(size_t)d * R + r]) / sg;
            w *= exp(-zz * zz / 2) * cnorm;
        }
        ws += w;
#pragma unroll
        for (int v = 0; v < GPSAT_GLUE_MAXVARS; ++v)
            if (v < nvars) {
                // a NaN prediction (failed tile) is left out of the weighted sum while its weight stays in the
                // denominator: what the reference's groupby(...).sum() does (it skips NaN), postprocessing.py:512-520
                const double x = vals[(size_t)v * R + r];
                if (x == x) acc[v] += w * x;
            }
    }
    ws = wave_sum(ws);
#pragma unroll
    for (int v = 0; v < GPSAT_GLUE_MAXVARS; ++v) {
        if (v < nvars) {
            const double s = wave_sum(acc[v]);
            if (lane == 0) out[(size_t)v * G + g] = s / ws;
        }
    }
}

hipError_t launch_smooth(int T, const double* x, const double* y, const double* vals, double lx, double ly, double* out,
                         hipStream_t stream) {
    hipLaunchKernelGGL(smooth_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, T, x, y, vals, lx, ly, out);
    return hipGetLastError();
}

hipError_t launch_glue(int G, int ndim, int nvars, long long R, const long long* seg, const double* pred, const double* xprt,
                       const double* vals, double sigma, const double* sigma_rows, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(glue_kernel, dim3((G + 3) / 4), dim3(256), 0, stream, G, ndim, nvars, R, seg, pred, xprt, vals, sigma, sigma_rows, out);
    return hipGetLastError();
}

}  // namespace gpsat
