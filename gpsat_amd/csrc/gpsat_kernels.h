// gpsat_kernels.h -- internal interface between the C ABI (gpsat_capi.cpp) and the gfx950 kernels.
#ifndef GPSAT_KERNELS_H
#define GPSAT_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gpsat {

// All pointers are DEVICE pointers.
struct KernelArgs {
    int T, kernel, optimiser, max_iter, max_ls, NBmax;
    double ftol, gtol, adam_lr;
    double noise_rel;             // relative objective resolution of the arithmetic (line-search failure at the noise floor)
    const long long* obs_off;     // [T+1]
    const long long* pred_off;    // [T+1]
    const double* theta0;         // [T*H]
    const double* lo;             // [T*H]
    const double* hi;             // [T*H]
    const unsigned char* trainable;  // [H]
    const float* X;               // [sumN*D]
    const float* y;               // [sumN]
    const float* Xs;              // [sumP*D]
    double* theta;                // [T*H]
    double* nll;                  // [T]
    double* grad;                 // [T*H] or nullptr
    int* status;                  // [T]
    int* n_eval;                  // [T]
    int* n_iter;                  // [T] optimiser iterations completed, or nullptr
    float* f_mean;                // [sumP]
    float* f_var;
    float* y_var;
    const int* order;             // [T] tile processing order (largest first)
    int* queue;                   // work-queue head (zeroed before launch)
    float* ws;                    // per-workgroup workspace
    size_t ws_stride;             // floats per workgroup
    unsigned long long* prof;     // [NW*16] cycle counters (diagnostic build only, else nullptr)
    const long long* cov_off;     // [T+1] element offsets into f_cov, or nullptr
    float* f_cov;                 // per tile P x P posterior covariance, or nullptr
    int PCmax;                    // max prediction chunks per tile (only used with f_cov)
    // time slicing of the optimisation (seg_cost = 0: every tile runs to completion from `queue`):
    // tiles are served from a ring; after ~seg_cost / NB^3 evaluations an unfinished tile's optimiser state is saved and the
    // tile goes to the back of the ring, so that all tiles of a homogeneous batch finish together instead of leaving a tail
    unsigned long long* ring;     // [ring_mask + 1] entries (sequence + 1) << 32 | resumed << 31 | tile; first T preset
    int* ring_ctl;                // [0] pop counter, [16] push counter (preset T), [32] unfinished tiles (preset T)
    unsigned* state;              // [T][state_words] saved optimiser state (the kernel's Shared struct)
    int ring_mask, state_words, seg_cost;
    // cooperative tiles (nullptr: off): one CoopCtl per workgroup (gpsat_coop.h), zeroed before the launch; workgroups that
    // find no tile left attach themselves to a running tile and pull groups of its sweep / gradient queues
    void* coop;                   // [grid] CoopCtl
    int* coop_live;               // tiles not finished yet (preset T): the helpers' exit condition
    // teams (fp64 kernels, gpsat_kernels_f64.hip): team_size workgroups run one tile together; [grid / team_size] TeamCtl of
    // 256 bytes, zeroed before the launch
    int team_size;
    void* team_ctl;
    int coop_min_nb;              // smallest tile (block columns) worth helping
    int coop_hdiv;                // helpers wanted per tile: NB / coop_hdiv (1..7)
    int coop_force;               // developer / tests: every evaluation of a helpable tile runs the cooperative code path, helped or not
    // diagnostic builds only (-DGPSAT_DUMP, scripts/e48_dump_compare.py): per tile the factor square, DinvT, z, alpha and the
    // log-determinant of its LAST evaluation, [T][dump_stride] floats in device memory; nullptr in the product
    float* dump;
    size_t dump_stride;
};

size_t shared_bytes(int D, int NBmax);
// fp64 kernels (gpsat_kernels_f64.hip): X, y, Xs, f_* and ws of KernelArgs point at doubles, ws_stride counts doubles
size_t shared_bytes_f64(int D, int NBmax);
size_t workspace_doubles_per_wg_f64(int NBmax, int PCcov);
int state_words_f64();
hipError_t launch_tiles_f64(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream);
// 4-wave build (gpsat_kernels_f64.hip -DGPSAT_F64_W4): two workgroups per CU for tiles whose LDS fits twice
size_t shared_bytes_f64_w4(int D, int NBmax);
size_t workspace_doubles_per_wg_f64_w4(int NBmax, int PCcov);
int state_words_f64_w4();
hipError_t launch_tiles_f64_w4(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream);
size_t workspace_floats_per_wg(int NBmax, int PCcov);     // PCcov: prediction chunks kept for f_cov (0 = none)
hipError_t launch_tiles(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream);
int state_words();                                        // 32-bit words of saved optimiser state per tile (time slicing)
// 8-wave build of the same kernels (gpsat_kernels.hip -DGPSAT_W8): used when a workgroup needs more than half of the LDS
size_t shared_bytes_w8(int D, int NBmax);
size_t workspace_floats_per_wg_w8(int NBmax, int PCcov);
hipError_t launch_tiles_w8(int D, const KernelArgs& a, int grid, size_t smem, hipStream_t stream);
int state_words_w8();

#define GPSAT_SEL_MAXCRIT 4

// tile selection (gpsat_select.hip); all pointers are device pointers
struct SelectArgs {
    int n_crit;
    int kind[GPSAT_SEL_MAXCRIT];      // 0: 1-D compare, 1: Euclidean ball
    int comp[GPSAT_SEL_MAXCRIT];      // 0 >=, 1 >, 2 ==, 3 <, 4 <=
    int ncols[GPSAT_SEL_MAXCRIT];
    int cols[GPSAT_SEL_MAXCRIT][3];
    double val[GPSAT_SEL_MAXCRIT];
    long long M;                      // rows of the point table
    int C;                            // columns of the point table / reference table
    int T;                            // experts
    const double* pts;                // [C][M] column-major (SoA)
    const double* refs;               // [T][C] row-major
    int n_chunks;                     // row chunks (grid.y)
    long long chunk_rows;             // rows per chunk (multiple of 64)
    long long* counts;                // [T][n_chunks]   (count pass)
    const long long* off;             // [T][n_chunks] start offsets (fill pass)
    int* idx;                         // [off[T]] (fill pass)
    const double* box;                // [ceil(M / sub)][C][2] per-column [min, max] of every sub-chunk of rows, or nullptr
    const int* eorder;                // [T] order in which the experts are dealt to the waves (neighbours together), or nullptr
};

hipError_t launch_select(const SelectArgs& a, bool fill, hipStream_t stream);
hipError_t launch_select_boxes(long long M, int C, const double* pts, double* box, hipStream_t stream);
int select_sub_rows();      // rows per box; chunk_rows must be a multiple of it

// Spatial binning of the point table (gpsat_select.hip): rows sorted by the cell of up to 3 columns, so that the boxes of
// consecutive rows are tight whatever order the table came in.
struct BinSpec {
    int ndim;                         // binned columns (1..3)
    int col[3];
    double origin[3], inv_cell[3];
    int ncell[3];
};
hipError_t select_bin_rows(long long M, int C, const double* pts, const BinSpec& b, unsigned* keys, unsigned* keys_out, int* rows,
                           int* perm, double* pts_perm, void* temp, size_t& temp_bytes, hipStream_t stream);
// selected positions of the binned table -> source rows, every expert's list ascending (the reference's source row order)
hipError_t select_unbin(int T, long long total, const unsigned* seg_off, const int* perm, int* idx, int* idx_out, void* temp,
                        size_t& temp_bytes, hipStream_t stream);

#define GPSAT_GLUE_MAXVARS 4
// post-processing (gpsat_post.hip); device pointers
hipError_t launch_smooth(int T, const double* x, const double* y, const double* vals, double lx, double ly, double* out,
                         hipStream_t stream);
hipError_t launch_glue(int G, int ndim, int nvars, long long R, const long long* seg, const double* pred, const double* xprt,
                       const double* vals, double sigma, const double* sigma_rows, double* out, hipStream_t stream);

}  // namespace gpsat
#endif
