// gpsat_capi.cpp -- C ABI of libgpsat_hip.so (see include/gpsat_hip.h for the contract and the
// reference interfaces each entry point replaces).  Host-side responsibilities only: argument
// validation, device buffers owned by the handle, cost-sorted tile order, launch, copy-back.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "gpsat_hip.h"
#include "gpsat_kernels.h"

#define GPSAT_PT_MAXNB_HOST 100        // colrow[] words in a CoopCtl (gpsat_coop.h: GPSAT_PT_MAXNB)

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? GPSAT_ENOMEM : GPSAT_EHIP,                    \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

// Developer knobs (GPSAT_DEBUG_*: grid size, slice length, cooperative-tile modes, team size, statistics) are read ONLY when
// GPSAT_DEVELOPER=1 is set in the environment: a stray GPSAT_DEBUG_* variable never changes what the shipped library does.
inline const char* dev_env(const char* name) {
    const char* d = std::getenv("GPSAT_DEVELOPER");
    if (!d || d[0] != '1' || d[1] != '\0') return nullptr;
    return std::getenv(name);
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return GPSAT_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(GPSAT_ENOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
        }
        cap = want;
        return GPSAT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// FNV-1a over all of `refs` and a sample of at most 65 536 evenly spaced elements of `points` (plus both ends): a caller who
// refills the same host buffers between the sizes call and the fill call gets a fresh selection, not the cached one
static unsigned long long sel_fingerprint(const double* points, long long nP, const double* refs, long long nR) {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const double* p) {
        unsigned long long v;
        std::memcpy(&v, p, 8);
        h = (h ^ v) * 1099511628211ull;
    };
    for (long long i = 0; i < nR; ++i) mix(refs + i);
    const long long step = std::max<long long>(1, nP / 65536);
    for (long long i = 0; i < nP; i += step) mix(points + i);
    for (long long i = std::max<long long>(0, nP - 64); i < nP; ++i) mix(points + i);
    return h;
}

struct gpsat_handle {
    int device = 0;
    int num_cu = 0;
    int wg_per_cu = 2;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    char name[256] = {0};
    double last_kernel_ms = 0.0, last_total_ms = 0.0;
    bool force_unsliced = false;       // retry of a batch whose time-sliced queue ended with unfinished tiles
    bool force_solo = false;           // retry of a batch in which a team barrier gave up
    // gpsat_select_batch is called twice per selection (sizes, then indices): the first call already leaves the indices on
    // the device; the second, when it repeats the first call's arguments, only copies them out
    struct {
        const void *pts = nullptr, *refs = nullptr;
        int64_t M = 0, total = -1;
        int C = 0, T = 0;
        gpsat_select_spec sp;
        unsigned long long fp = 0;     // fingerprint of the tables' CONTENTS at the sizes call
        const int* d_result = nullptr;
        std::vector<int64_t> off;
    } selc;
    // device buffers (grown lazily, owned by the handle)
    DevBuf meta_i64, meta_f64, meta_misc, out_f64, out_i32, bulk_in, bulk_out, ws, prof, ring, state, coop;
    DevBuf sel_pts, sel_refs, sel_cnt, sel_idx, sel_box, sel_perm, sel_keys, sel_tmp, sel_ord;
    float* dump_dev = nullptr;         // diagnostic build (-DGPSAT_DUMP): caller's device buffer for per-tile factor dumps
    size_t dump_stride = 0;
    unsigned long long prof_host[64 + 8 * 1024 + 2048] = {0};     // counters + event trace + per-workgroup start / end (diagnostic build)
};

extern "C" {

int gpsat_version(void) { return GPSAT_ABI_VERSION; }

const char* gpsat_last_error(void) { return g_err.c_str(); }

int gpsat_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gpsat_max_tile_obs(int dtype, int D) {
    // the largest tile whose workgroup state (coordinates, y, z, alpha, factor buffers, optimiser state) fits the
    // 160 KiB LDS of a CU with one workgroup per CU (8-wave builds)
    if (D < 1 || D > 4 || (dtype != GPSAT_F32 && dtype != GPSAT_F64)) return 0;
    const bool f64 = dtype == GPSAT_F64;
    const int bs = f64 ? 16 : 32;
    for (int NB = 4096 / bs; NB >= 1; --NB) {
        const size_t smem = f64 ? gpsat::shared_bytes_f64(D, NB) : gpsat::shared_bytes_w8(D, NB);
        if (smem <= 160 * 1024) return NB * bs;
    }
    return 0;
}

int gpsat_create(int device_id, const gpsat_opts* opts, gpsat_handle** out) {
    if (!out) return fail(GPSAT_EINVAL, "gpsat_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(GPSAT_ENODEV, "gpsat_create: no HIP device visible");
    if (device_id < 0 || device_id >= n) return fail(GPSAT_EINVAL, "gpsat_create: device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    gpsat_handle* h = new (std::nothrow) gpsat_handle();
    if (!h) return fail(GPSAT_ENOMEM, "gpsat_create: host allocation failed");
    h->device = device_id;
    h->num_cu = prop.multiProcessorCount;
    std::snprintf(h->name, sizeof(h->name), "%s (%s)", prop.name, prop.gcnArchName);
    if (opts && opts->workgroups_per_cu > 0) h->wg_per_cu = std::min(opts->workgroups_per_cu, 8);
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(GPSAT_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    for (int i = 0; i < 4; ++i) {
        e = hipEventCreate(&h->ev[i]);
        if (e != hipSuccess) { gpsat_destroy(h); return fail(GPSAT_EHIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    }
    *out = h;
    return GPSAT_OK;
}

int gpsat_device_name(gpsat_handle* h, char* buf, int buflen) {
    if (!h || !buf || buflen <= 0) return fail(GPSAT_EINVAL, "gpsat_device_name: bad argument");
    std::snprintf(buf, (size_t)buflen, "%s", h->name);
    return GPSAT_OK;
}

int gpsat_destroy(gpsat_handle* h) {
    if (!h) return GPSAT_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->meta_i64.release(); h->meta_f64.release(); h->meta_misc.release(); h->out_f64.release();
    h->out_i32.release(); h->bulk_in.release(); h->bulk_out.release(); h->ws.release(); h->prof.release(); h->ring.release(); h->state.release(); h->coop.release();
    h->sel_pts.release(); h->sel_refs.release(); h->sel_cnt.release(); h->sel_idx.release(); h->sel_box.release();
    h->sel_perm.release(); h->sel_keys.release(); h->sel_tmp.release(); h->sel_ord.release();
    for (int i = 0; i < 4; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GPSAT_OK;
}

int gpsat_last_timing(gpsat_handle* h, double* kernel_ms, double* total_ms) {
    if (!h) return fail(GPSAT_EINVAL, "gpsat_last_timing: handle is NULL");
    if (kernel_ms) *kernel_ms = h->last_kernel_ms;
    if (total_ms) *total_ms = h->last_total_ms;
    return GPSAT_OK;
}

int gpsat_fit_predict_batch(gpsat_handle* h, const gpsat_batch* b) {
    if (!h || !b) return fail(GPSAT_EINVAL, "gpsat_fit_predict_batch: NULL handle or batch");
    h->selc.total = -1;               // any other call on the handle ends a pending two-call selection
    if (b->T < 0) return fail(GPSAT_EINVAL, "T < 0");
    if (b->T == 0) return GPSAT_OK;
    if (b->D < 1 || b->D > 4) return fail(GPSAT_EINVAL, "D must be 1..4 in this build");
    if (b->dtype != GPSAT_F32 && b->dtype != GPSAT_F64) return fail(GPSAT_EINVAL, "unknown dtype");
    const bool f64 = b->dtype == GPSAT_F64;
    const size_t esz = f64 ? sizeof(double) : sizeof(float);
    if (b->kernel < 0 || b->kernel > 3) return fail(GPSAT_EINVAL, "unknown kernel id");
    if (b->optimiser < 0 || b->optimiser > 2) return fail(GPSAT_EINVAL, "unknown optimiser id");
    if (b->memory != GPSAT_MEM_HOST && b->memory != GPSAT_MEM_DEVICE) return fail(GPSAT_EINVAL, "bad memory flag");
    if (!b->obs_off || !b->pred_off || !b->theta0 || !b->lo || !b->hi || !b->trainable)
        return fail(GPSAT_EINVAL, "metadata pointer is NULL");
    if (!b->theta || !b->nll || !b->status || !b->n_eval) return fail(GPSAT_EINVAL, "output pointer is NULL");
    const int T = b->T, D = b->D, H = D + 2;
    // ---- validate CSR offsets, find the largest tile
    long long maxN = 0;
    if (b->obs_off[0] != 0 || b->pred_off[0] != 0) return fail(GPSAT_EINVAL, "offsets must start at 0");
    for (int t = 0; t < T; ++t) {
        const long long n = b->obs_off[t + 1] - b->obs_off[t], p = b->pred_off[t + 1] - b->pred_off[t];
        if (n < 0 || p < 0) return fail(GPSAT_EINVAL, "offsets must be non-decreasing");
        maxN = std::max(maxN, n);
    }
    const long long sumN = b->obs_off[T], sumP = b->pred_off[T];
    // optional full posterior covariance: one P_t x P_t block per tile
    const bool want_cov = b->f_cov != nullptr;
    long long sumC = 0, maxP = 0;
    if (want_cov) {
        if (!b->cov_off) return fail(GPSAT_EINVAL, "f_cov given without cov_off");
        if (b->cov_off[0] != 0) return fail(GPSAT_EINVAL, "cov_off must start at 0");
        for (int t = 0; t < T; ++t) {
            const long long p = b->pred_off[t + 1] - b->pred_off[t];
            if (b->cov_off[t + 1] - b->cov_off[t] != p * p) return fail(GPSAT_EINVAL, "cov_off[t+1]-cov_off[t] must equal P_t^2");
            maxP = std::max(maxP, p);
        }
        sumC = b->cov_off[T];
    }
    if (maxN > gpsat_max_tile_obs(b->dtype, b->D))
        return fail(GPSAT_EINVAL, "tile too large for the LDS of a CU: at most " + std::to_string(gpsat_max_tile_obs(b->dtype, b->D)) +
                                      " observations per tile for this dtype and D (gpsat_max_tile_obs)");
    if (sumN > 0 && (!b->X || !b->y)) return fail(GPSAT_EINVAL, "X / y is NULL");
    if (sumP > 0 && (!b->Xs || !b->f_mean || !b->f_var || !b->y_var)) return fail(GPSAT_EINVAL, "prediction pointer is NULL");
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < H; ++i) {
            const double v = b->theta0[(size_t)t * H + i];
            if (!(v > 0.0) || !std::isfinite(v)) return fail(GPSAT_EINVAL, "theta0 must be finite and positive");
        }
    const int bs = f64 ? 16 : 32;
    const int NBmax = std::max(1, (int)((maxN + bs - 1) / bs));

    HIP_TRY(hipSetDevice(h->device));
    // ---- tile order: largest cost first (N^3), stable so equal tiles keep the reference order
    std::vector<int> order(T);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
        return (b->obs_off[a + 1] - b->obs_off[a]) > (b->obs_off[c + 1] - b->obs_off[c]);
    });

    // ---- device buffers
    const size_t n_i64 = 3 * (size_t)(T + 1);
    const int PCcov = want_cov ? std::max(1, (int)((maxP + bs - 1) / bs)) : 0;
    const size_t n_f64 = 3 * (size_t)T * H;
    int rc;
    if ((rc = h->meta_i64.reserve(n_i64 * sizeof(long long)))) return rc;
    if ((rc = h->meta_f64.reserve(n_f64 * sizeof(double)))) return rc;
    if ((rc = h->meta_misc.reserve((size_t)T * sizeof(int) + 64 + 16))) return rc;
    if ((rc = h->out_f64.reserve(((size_t)T * H * 2 + (size_t)T) * sizeof(double)))) return rc;
    if ((rc = h->out_i32.reserve((size_t)T * 3 * sizeof(int)))) return rc;
    // fp32: two 4-wave workgroups per CU while a workgroup's LDS fits twice; beyond that one 8-wave workgroup per CU
    // (the 8-wave build of the same kernels), so that every SIMD still has two waves to overlap
    // ... and launches with fewer tiles than CUs: every tile has a CU to itself, eight waves use it better than four, and
    // the 8-wave build is the one with cooperative tiles
    const bool w8 = !f64 && (gpsat::shared_bytes(D, NBmax) > 80 * 1024 || h->wg_per_cu == 1 || T < h->num_cu);
    // fp64: the same rule with the 4-wave / 8-wave builds of the fp64 kernels
    const bool d4 = f64 && gpsat::shared_bytes_f64_w4(D, NBmax) <= 80 * 1024 && h->wg_per_cu != 1;
    const size_t wsf = f64 ? (d4 ? gpsat::workspace_doubles_per_wg_f64_w4(NBmax, PCcov) : gpsat::workspace_doubles_per_wg_f64(NBmax, PCcov))
                           : (w8 ? gpsat::workspace_floats_per_wg_w8(NBmax, PCcov) : gpsat::workspace_floats_per_wg(NBmax, PCcov));
    int grid = std::min(T, h->num_cu * (f64 ? 2 : h->wg_per_cu));
    const size_t smem = f64 ? (d4 ? gpsat::shared_bytes_f64_w4(D, NBmax) : gpsat::shared_bytes_f64(D, NBmax))
                            : (w8 ? gpsat::shared_bytes_w8(D, NBmax) : gpsat::shared_bytes(D, NBmax));
    if (smem > 160 * 1024) return fail(GPSAT_EINVAL, "tile too large for LDS");
    if (w8 || (f64 && !d4)) grid = std::min(grid, h->num_cu);
    // teams (fp64 kernels, 8-wave build): with few large tiles, G workgroups run every tile together (gpsat_kernels_f64.hip)
    int team = 1;
    if (f64 && !d4 && NBmax >= 64 && 2 * T <= h->num_cu) team = std::min(16, h->num_cu / T);
    if (const char* e = dev_env("GPSAT_DEBUG_TEAM")) { if (f64 && !d4) team = std::max(1, std::min(32, std::atoi(e))); }
    if (h->force_solo) team = 1;
    if (team > 1) grid = std::min(T, std::max(1, h->num_cu / team)) * team;
    // cooperative tiles (fp32 kernels): a workgroup without a tile helps a running one (gpsat_coop.h).  With fewer tiles than
    // resident workgroups the launch is widened by the helpers the large tiles can use.
    bool coop = !f64;
    int coop_min_nb = 12, coop_hdiv = 12;
    int coop_force = 0;
    if (const char* e = dev_env("GPSAT_DEBUG_COOP")) {           // developer: 0 = off, 2 = cooperative code path always
        coop = coop && std::atoi(e) != 0;
        coop_force = std::atoi(e) == 2;
    }
    if (const char* e = dev_env("GPSAT_DEBUG_COOP_XCD")) coop_force |= (std::atoi(e) & 3) << 2;   // developer: 1 same-XCD helpers only, 2 others only
    if (const char* e = dev_env("GPSAT_DEBUG_COOP_MIN_NB")) coop_min_nb = std::max(2, std::atoi(e));
    if (const char* e = dev_env("GPSAT_DEBUG_COOP_HDIV")) coop_hdiv = std::max(1, std::atoi(e));
    // Helpers must be capacity that would otherwise idle.  8-wave build: one workgroup per CU, a workgroup without a tile
    // leaves its CU empty -- always on.  4-wave build (two workgroups per CU): an idle workgroup's CU-mate already runs 1.4 x
    // faster alone, and a helper takes that back (measured on BASELINE configs[1]: the helped tail is 2 % SLOWER) -- off; a
    // launch with fewer tiles than CUs runs the 8-wave build anyway, widened to at most one workgroup per CU.
    if (coop && !w8) coop = false;
    if (coop) {
        const int cap = h->num_cu;
        long long want = grid;
        for (int t = 0; t < T && want < cap; ++t) {
            const int nb = (int)((b->obs_off[t + 1] - b->obs_off[t] + bs - 1) / bs);
            if (nb >= coop_min_nb) want += std::min(7, std::max(1, nb / coop_hdiv));
        }
        if (T < cap) grid = (int)std::min<long long>(cap, want);
    }
    if (const char* e = dev_env("GPSAT_DEBUG_GRID")) grid = std::max(1, std::min(grid, std::atoi(e)));   // developer: fewer resident workgroups
    if ((rc = h->ws.reserve((size_t)(grid / team) * wsf * esz))) return rc;      // one workspace per workgroup, or per team

    const char *dX = nullptr, *dy = nullptr, *dXs = nullptr;
    char *dfm = nullptr, *dfv = nullptr, *dyv = nullptr, *dcov = nullptr;
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    if (b->memory == GPSAT_MEM_HOST) {
        const size_t in_e = (size_t)sumN * D + (size_t)sumN + (size_t)sumP * D;
        if ((rc = h->bulk_in.reserve(std::max<size_t>(in_e, 1) * esz))) return rc;
        if ((rc = h->bulk_out.reserve(std::max<size_t>((size_t)sumP * 3 + (size_t)sumC, 1) * esz))) return rc;
        char* base = static_cast<char*>(h->bulk_in.p);
        dX = base; dy = base + (size_t)sumN * D * esz; dXs = base + (size_t)sumN * (D + 1) * esz;
        if (sumN > 0) {
            HIP_TRY(hipMemcpyAsync(const_cast<char*>(dX), b->X, (size_t)sumN * D * esz, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(const_cast<char*>(dy), b->y, (size_t)sumN * esz, hipMemcpyHostToDevice, h->stream));
        }
        if (sumP > 0)
            HIP_TRY(hipMemcpyAsync(const_cast<char*>(dXs), b->Xs, (size_t)sumP * D * esz, hipMemcpyHostToDevice, h->stream));
        dfm = static_cast<char*>(h->bulk_out.p); dfv = dfm + (size_t)sumP * esz; dyv = dfv + (size_t)sumP * esz;
        if (want_cov) dcov = dyv + (size_t)sumP * esz;
    } else {
        dcov = static_cast<char*>(b->f_cov);
        dX = static_cast<const char*>(b->X); dy = static_cast<const char*>(b->y); dXs = static_cast<const char*>(b->Xs);
        dfm = static_cast<char*>(b->f_mean); dfv = static_cast<char*>(b->f_var); dyv = static_cast<char*>(b->y_var);
    }
    long long* d_i64 = static_cast<long long*>(h->meta_i64.p);
    HIP_TRY(hipMemcpyAsync(d_i64, b->obs_off, (size_t)(T + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_i64 + (T + 1), b->pred_off, (size_t)(T + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    if (want_cov)
        HIP_TRY(hipMemcpyAsync(d_i64 + 2 * (T + 1), b->cov_off, (size_t)(T + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    double* d_f64 = static_cast<double*>(h->meta_f64.p);
    HIP_TRY(hipMemcpyAsync(d_f64, b->theta0, (size_t)T * H * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_f64 + (size_t)T * H, b->lo, (size_t)T * H * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_f64 + 2 * (size_t)T * H, b->hi, (size_t)T * H * sizeof(double), hipMemcpyHostToDevice, h->stream));
    unsigned char* d_misc = static_cast<unsigned char*>(h->meta_misc.p);
    int* d_queue = reinterpret_cast<int*>(d_misc);            // 16 bytes reserved
    unsigned char* d_train = d_misc + 16;                     // 64 bytes reserved
    int* d_order = reinterpret_cast<int*>(d_misc + 16 + 64);
    HIP_TRY(hipMemsetAsync(d_queue, 0, 16, h->stream));
    HIP_TRY(hipMemcpyAsync(d_train, b->trainable, (size_t)H, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, h->stream));

    // ---- time slicing of the optimisation (fp32): with few tiles per resident workgroup, whole tiles as the scheduling unit
    // leave the GPU half empty while the last ones finish (4096 tiles on 512 workgroups: 8 % of the launch).  Tiles of
    // similar cost are therefore served in slices of ~4 evaluations of a 512-point tile (both precisions); a batch whose largest tile
    // dominates keeps the largest-first run-to-completion order (its critical path must not wait in a queue).
    int seg_cost = 0;
    if (b->optimiser != GPSAT_OPT_NONE && b->max_iter > 0) {
        double sum_cost = 0.0, max_cost = 0.0;
        for (int t = 0; t < T; ++t) {
            const double nb = (double)((b->obs_off[t + 1] - b->obs_off[t] + bs - 1) / bs);
            sum_cost += nb * nb * nb;
            max_cost = std::max(max_cost, nb * nb * nb);
        }
        const double tiles_per_wg = (double)T / grid;
        if (T > grid && max_cost * 4.0 * grid <= sum_cost && tiles_per_wg <= 64.0) seg_cost = 4 * (512 / bs) * (512 / bs) * (512 / bs);
        // developer / tests: slice length in NB^3 units (0 = off, 1 = every evaluation), whatever the batch looks like
        if (const char* e = dev_env("GPSAT_DEBUG_SEG")) seg_cost = std::max(0, std::atoi(e));
        if (h->force_unsliced || team > 1) seg_cost = 0;
    }
    unsigned long long* d_ring = nullptr; int* d_ring_ctl = nullptr; unsigned* d_state = nullptr;
    int ring_mask = 0;
    const int state_words = f64 ? (d4 ? gpsat::state_words_f64_w4() : gpsat::state_words_f64())
                                : (w8 ? gpsat::state_words_w8() : gpsat::state_words());
    if (seg_cost > 0) {
        size_t cap = 1; while (cap < (size_t)T + (size_t)grid + 1) cap <<= 1;
        ring_mask = (int)(cap - 1);
        if ((rc = h->ring.reserve(cap * sizeof(unsigned long long) + 256))) return rc;
        if ((rc = h->state.reserve((size_t)T * state_words * sizeof(unsigned)))) return rc;
        d_ring_ctl = static_cast<int*>(h->ring.p);
        d_ring = reinterpret_cast<unsigned long long*>(static_cast<char*>(h->ring.p) + 256);
        d_state = static_cast<unsigned*>(h->state.p);
        std::vector<unsigned long long> init(cap, 0ull);
        for (int i = 0; i < T; ++i) init[i] = ((unsigned long long)(i + 1) << 32) | (unsigned)order[i];
        int ctl[64] = {0};
        ctl[16] = T; ctl[32] = T;
        HIP_TRY(hipMemcpyAsync(d_ring_ctl, ctl, sizeof(ctl), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d_ring, init.data(), cap * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));          // `init` and `ctl` are stack / local host memory
    }

    gpsat::KernelArgs a;
    a.coop = nullptr; a.coop_live = nullptr; a.coop_min_nb = coop_min_nb; a.coop_hdiv = coop_hdiv; a.coop_force = coop_force;
    if (coop) {
        // [grid] control blocks of 1 KiB, zeroed every launch, then the count of unfinished tiles
        const size_t cb = (size_t)grid * 1024;
        if ((rc = h->coop.reserve(cb + 64))) return rc;
        HIP_TRY(hipMemsetAsync(h->coop.p, 0, cb + 64, h->stream));
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(h->coop.p) + cb, &b->T, sizeof(int), hipMemcpyHostToDevice, h->stream));
        a.coop = h->coop.p;
        a.coop_live = reinterpret_cast<int*>(static_cast<char*>(h->coop.p) + cb);
    }
    a.team_size = team; a.team_ctl = nullptr;
    if (team > 1) {
        const size_t tb = (size_t)(grid / team) * 256;
        if ((rc = h->coop.reserve(tb))) return rc;
        HIP_TRY(hipMemsetAsync(h->coop.p, 0, tb, h->stream));
        a.team_ctl = h->coop.p;
    }
    a.ring = d_ring; a.ring_ctl = d_ring_ctl; a.state = d_state; a.ring_mask = ring_mask; a.state_words = state_words; a.seg_cost = seg_cost;
    a.T = T; a.kernel = b->kernel; a.optimiser = b->optimiser; a.max_iter = b->max_iter;
    a.max_ls = b->max_ls > 0 ? b->max_ls : 20;                                 // SciPy L-BFGS-B maxls
    a.NBmax = NBmax;
    // 0 = default (fp64: SciPy's factr*eps; fp32: its analogue above the fp32 noise floor); negative = criterion off
    a.ftol = b->ftol > 0 ? b->ftol : (b->ftol < 0 ? -1.0 : (f64 ? 2.220446049250313e-9 : 1e-6));
    a.gtol = b->gtol > 0 ? b->gtol : (b->gtol < 0 ? -1.0 : 1e-5);
    a.adam_lr = b->adam_lr > 0 ? b->adam_lr : 0.1;
    // relative resolution of the objective: fp32 rounding ~ cond(K) eps N reaches 2e-4 |f| on the reference's 1-D tutorial
    // tile (cond ~ 2e4, docs/notebooks/1d_local_expert_model_part_2.ipynb); fp64: rounding level only
    a.noise_rel = f64 ? 1e-12 : 1e-3;
    a.obs_off = d_i64; a.pred_off = d_i64 + (T + 1);
    a.theta0 = d_f64; a.lo = d_f64 + (size_t)T * H; a.hi = d_f64 + 2 * (size_t)T * H;
    a.trainable = d_train;
    a.X = reinterpret_cast<const float*>(dX); a.y = reinterpret_cast<const float*>(dy); a.Xs = reinterpret_cast<const float*>(dXs);
    double* d_out = static_cast<double*>(h->out_f64.p);
    a.theta = d_out; a.nll = d_out + (size_t)T * H;
    a.grad = b->grad ? d_out + (size_t)T * H + T : nullptr;
    int* d_oi = static_cast<int*>(h->out_i32.p);
    a.status = d_oi; a.n_eval = d_oi + T; a.n_iter = d_oi + 2 * (size_t)T;
    a.f_mean = reinterpret_cast<float*>(dfm); a.f_var = reinterpret_cast<float*>(dfv); a.y_var = reinterpret_cast<float*>(dyv);
    a.order = d_order; a.queue = d_queue;
    a.ws = static_cast<float*>(h->ws.p); a.ws_stride = wsf;     // fp64: the kernel reinterprets ws as doubles
    a.prof = nullptr;
    a.cov_off = want_cov ? d_i64 + 2 * (T + 1) : nullptr;
    a.f_cov = want_cov ? reinterpret_cast<float*>(dcov) : nullptr;
    a.PCmax = PCcov;
    a.dump = nullptr; a.dump_stride = 0;
#ifdef GPSAT_DUMP
    if (h->dump_dev && !f64) {
        const size_t need = ((size_t)NBmax * NBmax + NBmax) * 1024 + 2 * (size_t)NBmax * 32 + 16 + 8 * 1024;
        if (h->dump_stride < need) return fail(GPSAT_EINVAL, "dump stride too small: need " + std::to_string(need) + " floats per tile");
        a.dump = h->dump_dev; a.dump_stride = h->dump_stride;
    }
#endif
#ifdef GPSAT_PROFILE
    if ((rc = h->prof.reserve(sizeof(h->prof_host)))) return rc;
    HIP_TRY(hipMemsetAsync(h->prof.p, 0, sizeof(h->prof_host), h->stream));
    a.prof = static_cast<unsigned long long*>(h->prof.p);
#endif

    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    HIP_TRY(f64 ? (d4 ? gpsat::launch_tiles_f64_w4(D, a, grid, smem, h->stream) : gpsat::launch_tiles_f64(D, a, grid, smem, h->stream))
                : (w8 ? gpsat::launch_tiles_w8(D, a, grid, smem, h->stream) : gpsat::launch_tiles(D, a, grid, smem, h->stream)));
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));

    HIP_TRY(hipMemcpyAsync(b->theta, a.theta, (size_t)T * H * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(b->nll, a.nll, (size_t)T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (b->grad) HIP_TRY(hipMemcpyAsync(b->grad, a.grad, (size_t)T * H * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(b->status, a.status, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(b->n_eval, a.n_eval, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (b->n_iter) HIP_TRY(hipMemcpyAsync(b->n_iter, a.n_iter, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (b->memory == GPSAT_MEM_HOST && sumP > 0) {
        HIP_TRY(hipMemcpyAsync(b->f_mean, dfm, (size_t)sumP * esz, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(b->f_var, dfv, (size_t)sumP * esz, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(b->y_var, dyv, (size_t)sumP * esz, hipMemcpyDeviceToHost, h->stream));
    }
    if (b->memory == GPSAT_MEM_HOST && want_cov && sumC > 0)
        HIP_TRY(hipMemcpyAsync(b->f_cov, dcov, (size_t)sumC * esz, hipMemcpyDeviceToHost, h->stream));
#ifdef GPSAT_PROFILE
    HIP_TRY(hipMemcpyAsync(h->prof_host, h->prof.p, sizeof(h->prof_host), hipMemcpyDeviceToHost, h->stream));
#endif
    std::vector<int> team_host;
    if (team > 1) {
        team_host.resize((size_t)(grid / team) * 64);
        HIP_TRY(hipMemcpyAsync(team_host.data(), h->coop.p, team_host.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    std::vector<int> coop_host;
    if (coop && dev_env("GPSAT_DEBUG_COOP_STATS")) {
        coop_host.resize((size_t)grid * 256);
        HIP_TRY(hipMemcpyAsync(coop_host.data(), h->coop.p, (size_t)grid * 1024, hipMemcpyDeviceToHost, h->stream));
    }
    int unfinished = 0;
    if (seg_cost > 0) HIP_TRY(hipMemcpyAsync(&unfinished, d_ring_ctl + 32, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!coop_host.empty()) {
        long long st[8] = {0};
        for (int g = 0; g < grid; ++g) for (int i = 0; i < 8; ++i) st[i] += coop_host[(size_t)g * 256 + 32 + GPSAT_PT_MAXNB_HOST + i];
        std::fprintf(stderr, "gpsat coop: grid %d T %d: cooperative evaluations %lld, helper phases %lld, helper groups (sweep) %lld, "
                             "flag waits given up %lld, owner waits given up %lld, pivot failures %lld, helper unwinds %lld\n",
                     grid, T, st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
    }
    if (!team_host.empty() && dev_env("GPSAT_DEBUG_TEAM_STATS"))
        std::fprintf(stderr, "gpsat team 0 (size %d), factorisation, owner thread 0, s_memtime ticks: own work of (A) %d, (A) wait + barrier %d, (B) + barrier %d, "
                             "(C) + barrier %d\n", team, team_host[24], team_host[25], team_host[26], team_host[27]);
    for (size_t g = 0; g < team_host.size() / 64; ++g) {
        if (team_host[g * 64 + 5]) {          // TeamCtl::timeout: a team barrier gave up (never by design) -- run the batch again, one workgroup per tile
            std::fprintf(stderr, "gpsat: a team barrier gave up; re-running the batch with one workgroup per tile\n");
            h->force_solo = true;
            const int rc2 = gpsat_fit_predict_batch(h, b);
            h->force_solo = false;
            return rc2;
        }
    }
    if (unfinished != 0) {
        // A queue anomaly (an escape hatch of ring_pop taken: gpsat_ring.h) must not cost the caller the batch: run it again
        // with every tile run to completion from the plain queue (same results: slicing does not change a bit of them).
        if (h->force_unsliced) return fail(GPSAT_EHIP, "tile queue ended with " + std::to_string(unfinished) + " unfinished tiles");
        std::fprintf(stderr, "gpsat: time-sliced tile queue ended with %d unfinished tiles; re-running the batch unsliced\n", unfinished);
        h->force_unsliced = true;
        const int rc2 = gpsat_fit_predict_batch(h, b);
        h->force_unsliced = false;
        return rc2;
    }
    float km = 0.f, tm = 0.f;
    HIP_TRY(hipEventElapsedTime(&km, h->ev[1], h->ev[2]));
    HIP_TRY(hipEventElapsedTime(&tm, h->ev[0], h->ev[3]));
    h->last_kernel_ms = km;
    h->last_total_ms = tm;
    return GPSAT_OK;
}

#ifdef GPSAT_DUMP
// diagnostic build only: device buffer [T][stride_floats] that every fp32 batch call fills (KernelArgs::dump)
int gpsat_debug_set_dump(gpsat_handle* h, void* dev, unsigned long long stride_floats) {
    if (!h) return GPSAT_EINVAL;
    h->dump_dev = static_cast<float*>(dev); h->dump_stride = (size_t)stride_floats;
    return GPSAT_OK;
}
#endif

int gpsat_select_batch(gpsat_handle* h, const gpsat_select_spec* sp, int64_t M, int32_t C, const double* points,
                       int32_t T, const double* refs, int64_t* off, int32_t* idx, int64_t capacity) {
    if (!h || !sp || !off) return fail(GPSAT_EINVAL, "gpsat_select_batch: NULL argument");
    if (T < 0 || M < 0 || C < 1) return fail(GPSAT_EINVAL, "gpsat_select_batch: bad sizes");
    if (M > 2147483647LL) return fail(GPSAT_EINVAL, "gpsat_select_batch: more than 2^31-1 rows");
    if (sp->n_crit < 1 || sp->n_crit > GPSAT_SEL_MAXCRIT) return fail(GPSAT_EINVAL, "gpsat_select_batch: n_crit out of range");
    gpsat::SelectArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_crit = sp->n_crit;
    for (int k = 0; k < sp->n_crit; ++k) {
        if (sp->kind[k] != 0 && sp->kind[k] != 1) return fail(GPSAT_EINVAL, "gpsat_select_batch: bad criterion kind");
        if (sp->comp[k] < 0 || sp->comp[k] > 4) return fail(GPSAT_EINVAL, "gpsat_select_batch: bad comparison");
        const int nc = sp->kind[k] == 0 ? 1 : sp->ncols[k];
        if (nc < 1 || nc > 3) return fail(GPSAT_EINVAL, "gpsat_select_batch: ball criteria take 1..3 columns");
        if (sp->kind[k] == 1 && sp->comp[k] != 3 && sp->comp[k] != 4) return fail(GPSAT_EINVAL, "gpsat_select_batch: ball criteria are < or <=");
        a.kind[k] = sp->kind[k]; a.comp[k] = sp->comp[k]; a.ncols[k] = nc; a.val[k] = sp->val[k];
        for (int m = 0; m < nc; ++m) {
            if (sp->cols[k][m] < 0 || sp->cols[k][m] >= C) return fail(GPSAT_EINVAL, "gpsat_select_batch: column index out of range");
            a.cols[k][m] = sp->cols[k][m];
        }
    }
    off[0] = 0;
    if (T == 0) return GPSAT_OK;
    if ((M > 0 && !points) || !refs) return fail(GPSAT_EINVAL, "gpsat_select_batch: NULL table");
    if (idx && h->selc.total >= 0 && h->selc.pts == points && h->selc.refs == refs && h->selc.M == M && h->selc.C == C &&
        h->selc.T == T && std::memcmp(&h->selc.sp, sp, sizeof(*sp)) == 0 &&
        h->selc.fp == sel_fingerprint(points, (long long)M * C, refs, (long long)T * C)) {
        const int64_t total = h->selc.total;
        h->selc.total = -1;
        std::memcpy(off, h->selc.off.data(), (size_t)(T + 1) * sizeof(int64_t));
        if (capacity < total) return fail(GPSAT_EINVAL, "gpsat_select_batch: idx capacity too small (see off[T])");
        HIP_TRY(hipSetDevice(h->device));
        if (total > 0) HIP_TRY(hipMemcpy(idx, h->selc.d_result, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
        return GPSAT_OK;
    }
    h->selc.total = -1;
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = h->sel_pts.reserve(std::max<size_t>((size_t)M * C, 1) * sizeof(double)))) return rc;
    if ((rc = h->sel_refs.reserve((size_t)T * C * sizeof(double)))) return rc;
    // row chunks: enough workgroups to fill the chip (T/32 workgroups per chunk), chunk a multiple of 64 rows
    const int wgs_per_chunk = std::max(1, (T + 31) / 32);
    int n_chunks = (int)std::min<long long>(std::max<long long>(1, (4096 + wgs_per_chunk - 1) / wgs_per_chunk), std::max<long long>(1, (M + 4095) / 4096));
    const long long sub = gpsat::select_sub_rows();             // rows per bounding box: chunks are whole numbers of them
    long long chunk_rows = ((M + n_chunks - 1) / n_chunks + sub - 1) / sub * sub;
    if (chunk_rows < sub) chunk_rows = sub;
    n_chunks = (int)std::max<long long>(1, (M + chunk_rows - 1) / chunk_rows);
    const size_t ncell = (size_t)T * n_chunks;
    if ((rc = h->sel_cnt.reserve(2 * ncell * sizeof(long long)))) return rc;
    if (M > 0) HIP_TRY(hipMemcpyAsync(h->sel_pts.p, points, (size_t)M * C * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->sel_refs.p, refs, (size_t)T * C * sizeof(double), hipMemcpyHostToDevice, h->stream));
    a.M = M; a.C = C; a.T = T;
    a.n_chunks = n_chunks; a.chunk_rows = chunk_rows;
    a.eorder = nullptr;
    a.pts = static_cast<const double*>(h->sel_pts.p);
    // ---- spatial binning (gpsat_select.hip): large tables are sorted on the device by the grid cell of the criteria's
    // columns -- a ball criterion's columns with cells of its radius, a two-sided 1-D window's column with cells of half its
    // width, at most three columns -- and the experts are dealt to the waves in the order of their own cells
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    gpsat::BinSpec bin;
    bin.ndim = 0;
    const int* d_perm = nullptr;
    if (M >= 65536 && !dev_env("GPSAT_DEBUG_NO_BINNING")) {
        auto add_dim = [&](int col, double cell) {
            if (bin.ndim >= 3 || !(cell > 0.0) || !std::isfinite(cell)) return;
            for (int d = 0; d < bin.ndim; ++d) if (bin.col[d] == col) return;
            double mn = INFINITY, mx = -INFINITY;
            const double* x = points + (size_t)col * M;
            for (int64_t i = 0; i < M; ++i) { const double v = x[i]; if (v < mn) mn = v; if (v > mx) mx = v; }
            if (!(mn <= mx) || !std::isfinite(mn) || !std::isfinite(mx)) return;
            const double nc = std::min(1024.0, std::max(1.0, std::ceil((mx - mn) / cell)));
            bin.col[bin.ndim] = col; bin.origin[bin.ndim] = mn; bin.ncell[bin.ndim] = (int)nc;
            bin.inv_cell[bin.ndim] = (mx > mn) ? nc / (mx - mn) : 0.0;
            ++bin.ndim;
        };
        for (int k = 0; k < a.n_crit; ++k) {              // two-sided windows first (GPSat: the time column)
            if (a.kind[k] != 0 || !(a.comp[k] == 3 || a.comp[k] == 4)) continue;
            for (int k2 = 0; k2 < a.n_crit; ++k2)
                if (a.kind[k2] == 0 && (a.comp[k2] == 0 || a.comp[k2] == 1) && a.cols[k2][0] == a.cols[k][0] && a.val[k] > a.val[k2])
                    add_dim(a.cols[k][0], 0.5 * (a.val[k] - a.val[k2]));
        }
        for (int k = 0; k < a.n_crit; ++k)
            if (a.kind[k] == 1) for (int m = 0; m < a.ncols[k]; ++m) add_dim(a.cols[k][m], a.val[k]);
    }
    if (bin.ndim > 0) {
        const size_t perm_bytes = ((size_t)M * 2 * sizeof(int) + 255) & ~size_t(255);
        if ((rc = h->sel_perm.reserve(perm_bytes + (size_t)M * C * sizeof(double)))) return rc;
        if ((rc = h->sel_keys.reserve((size_t)M * 2 * sizeof(unsigned)))) return rc;
        int* d_rows = static_cast<int*>(h->sel_perm.p);
        int* d_p = d_rows + M;
        double* d_pp = reinterpret_cast<double*>(static_cast<char*>(h->sel_perm.p) + perm_bytes);
        unsigned* d_k = static_cast<unsigned*>(h->sel_keys.p);
        size_t tb = 0;
        HIP_TRY(gpsat::select_bin_rows(M, C, a.pts, bin, d_k, d_k + M, d_rows, d_p, d_pp, nullptr, tb, h->stream));
        if ((rc = h->sel_tmp.reserve(std::max<size_t>(tb, 16)))) return rc;
        HIP_TRY(gpsat::select_bin_rows(M, C, a.pts, bin, d_k, d_k + M, d_rows, d_p, d_pp, h->sel_tmp.p, tb, h->stream));
        a.pts = d_pp;
        d_perm = d_p;
        // experts by their own cell
        std::vector<unsigned> ekey(T);
        for (int t = 0; t < T; ++t) {
            unsigned key = 0;
            for (int d = 0; d < bin.ndim; ++d) {
                const double cf = (refs[(size_t)t * C + bin.col[d]] - bin.origin[d]) * bin.inv_cell[d];
                const int cell = (cf >= 0.0) ? (int)std::min(cf, (double)(bin.ncell[d] - 1)) : 0;
                key = key * (unsigned)bin.ncell[d] + (unsigned)cell;
            }
            ekey[t] = key;
        }
        std::vector<int> eord(T);
        std::iota(eord.begin(), eord.end(), 0);
        std::stable_sort(eord.begin(), eord.end(), [&](int x, int y) { return ekey[x] < ekey[y]; });
        if ((rc = h->sel_ord.reserve((size_t)T * sizeof(int) + (size_t)(T + 1) * sizeof(unsigned)))) return rc;
        HIP_TRY(hipMemcpyAsync(h->sel_ord.p, eord.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));          // `eord` is local host memory
        a.eorder = static_cast<const int*>(h->sel_ord.p);
    }
    a.refs = static_cast<const double*>(h->sel_refs.p);
    a.counts = static_cast<long long*>(h->sel_cnt.p);
    a.box = nullptr;
    if (M > 0) {
        // per-column [min, max] of every `sub` rows: lets a wave skip sub-chunks none of its experts can select from
        const size_t nsub = (size_t)((M + sub - 1) / sub);
        if ((rc = h->sel_box.reserve(nsub * C * 2 * sizeof(double)))) return rc;
        HIP_TRY(gpsat::launch_select_boxes(M, C, a.pts, static_cast<double*>(h->sel_box.p), h->stream));
        a.box = static_cast<const double*>(h->sel_box.p);
    }
    HIP_TRY(gpsat::launch_select(a, false, h->stream));
    std::vector<long long> cnt(ncell);
    HIP_TRY(hipMemcpyAsync(cnt.data(), a.counts, ncell * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // exclusive scan over (expert, chunk) cells, expert-major: cnt becomes the start offset of every cell
    long long run = 0;
    for (int t = 0; t < T; ++t) {
        off[t] = run;
        for (int cc = 0; cc < n_chunks; ++cc) { const long long v = cnt[(size_t)t * n_chunks + cc]; cnt[(size_t)t * n_chunks + cc] = run; run += v; }
    }
    off[T] = run;
    if (idx && capacity < off[T]) return fail(GPSAT_EINVAL, "gpsat_select_batch: idx capacity too small (see off[T])");
    const int* d_final = nullptr;
    if (off[T] > 0) {
        if ((rc = h->sel_idx.reserve((size_t)off[T] * sizeof(int)))) return rc;
        long long* d_off = static_cast<long long*>(h->sel_cnt.p) + ncell;
        HIP_TRY(hipMemcpyAsync(d_off, cnt.data(), ncell * sizeof(long long), hipMemcpyHostToDevice, h->stream));
        a.off = d_off;
        a.idx = static_cast<int*>(h->sel_idx.p);
        HIP_TRY(gpsat::launch_select(a, true, h->stream));
        const int* d_result = a.idx;
        if (d_perm) {
            // positions of the binned table -> source rows, every expert's list ascending (source row order)
            if (off[T] > 2147483647LL) return fail(GPSAT_EINVAL, "gpsat_select_batch: more than 2^31-1 selected rows");
            std::vector<unsigned> off32(T + 1);
            for (int t = 0; t <= T; ++t) off32[t] = (unsigned)off[t];
            unsigned* d_off32 = reinterpret_cast<unsigned*>(static_cast<char*>(h->sel_ord.p) + (size_t)T * sizeof(int));
            HIP_TRY(hipMemcpyAsync(d_off32, off32.data(), (size_t)(T + 1) * sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            if ((rc = h->sel_keys.reserve(std::max((size_t)M * 2 * sizeof(unsigned), (size_t)off[T] * sizeof(int))))) return rc;
            int* d_sorted = static_cast<int*>(h->sel_keys.p);            // the row keys are no longer needed
            size_t tb = 0;
            HIP_TRY(gpsat::select_unbin(T, off[T], d_off32, d_perm, a.idx, d_sorted, nullptr, tb, h->stream));
            if ((rc = h->sel_tmp.reserve(std::max<size_t>(tb, 16)))) return rc;
            HIP_TRY(gpsat::select_unbin(T, off[T], d_off32, d_perm, a.idx, d_sorted, h->sel_tmp.p, tb, h->stream));
            d_result = d_sorted;
        }
        HIP_TRY(hipEventRecord(h->ev[2], h->stream));
        d_final = d_result;
        if (idx) HIP_TRY(hipMemcpyAsync(idx, d_result, (size_t)off[T] * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    } else {
        HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    }
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float km = 0.f, tm = 0.f;
    HIP_TRY(hipEventElapsedTime(&km, h->ev[1], h->ev[2]));      // count + scan round trip + fill
    HIP_TRY(hipEventElapsedTime(&tm, h->ev[0], h->ev[3]));
    h->last_kernel_ms = km;
    h->last_total_ms = tm;
    if (!idx) {
        // sizes asked for: the indices stay on the device for the call that follows with the same arguments
        h->selc.pts = points; h->selc.refs = refs; h->selc.M = M; h->selc.C = C; h->selc.T = T; h->selc.sp = *sp;
        h->selc.d_result = d_final; h->selc.total = off[T];
        h->selc.fp = sel_fingerprint(points, (long long)M * C, refs, (long long)T * C);
        h->selc.off.assign(off, off + T + 1);
    }
    return GPSAT_OK;
}

int gpsat_smooth_batch(gpsat_handle* h, int32_t T, const double* x, const double* y, const double* vals, double l_x,
                       double l_y, double* out) {
    if (!h || T < 0 || (T > 0 && (!x || !y || !vals || !out))) return fail(GPSAT_EINVAL, "gpsat_smooth_batch: bad argument");
    if (!(l_x > 0.0) || !(l_y > 0.0)) return fail(GPSAT_EINVAL, "gpsat_smooth_batch: length scales must be positive");
    h->selc.total = -1;
    if (T == 0) return GPSAT_OK;
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = h->sel_pts.reserve((size_t)4 * T * sizeof(double)))) return rc;
    double* d = static_cast<double*>(h->sel_pts.p);
    HIP_TRY(hipMemcpyAsync(d, x, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d + T, y, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d + 2 * (size_t)T, vals, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    HIP_TRY(gpsat::launch_smooth(T, d, d + T, d + 2 * (size_t)T, l_x, l_y, d + 3 * (size_t)T, h->stream));
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipMemcpyAsync(out, d + 3 * (size_t)T, (size_t)T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float km = 0.f;
    HIP_TRY(hipEventElapsedTime(&km, h->ev[1], h->ev[2]));
    h->last_kernel_ms = km; h->last_total_ms = km;
    return GPSAT_OK;
}

int gpsat_glue_batch(gpsat_handle* h, int64_t R, int32_t G, int32_t ndim, int32_t nvars, const int64_t* seg,
                     const double* pred, const double* xprt, const double* vals, double sigma,
                     const double* sigma_rows, double* out) {
    if (!h || R < 0 || G < 0) return fail(GPSAT_EINVAL, "gpsat_glue_batch: bad sizes");
    if (ndim < 1 || ndim > 2 || nvars < 1 || nvars > GPSAT_GLUE_MAXVARS) return fail(GPSAT_EINVAL, "gpsat_glue_batch: ndim 1..2, nvars 1..4");
    if (!sigma_rows && !(sigma > 0.0)) return fail(GPSAT_EINVAL, "gpsat_glue_batch: sigma must be positive");
    if (G == 0) return GPSAT_OK;
    if (!seg || !pred || !xprt || !vals || !out) return fail(GPSAT_EINVAL, "gpsat_glue_batch: NULL argument");
    if (seg[0] != 0 || seg[G] != R) return fail(GPSAT_EINVAL, "gpsat_glue_batch: seg must run from 0 to R");
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    const size_t nd = (size_t)(2 * ndim + nvars + 1) * R + (size_t)nvars * G;
    if ((rc = h->sel_pts.reserve(std::max<size_t>(nd, 1) * sizeof(double)))) return rc;
    if ((rc = h->sel_cnt.reserve((size_t)(G + 1) * sizeof(long long)))) return rc;
    double* d = static_cast<double*>(h->sel_pts.p);
    double* dp = d; double* dx = d + (size_t)ndim * R; double* dv = dx + (size_t)ndim * R; double* dsig = dv + (size_t)nvars * R; double* dout = dsig + R;
    HIP_TRY(hipMemcpyAsync(dp, pred, (size_t)ndim * R * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dx, xprt, (size_t)ndim * R * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dv, vals, (size_t)nvars * R * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (sigma_rows) HIP_TRY(hipMemcpyAsync(dsig, sigma_rows, (size_t)R * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->sel_cnt.p, seg, (size_t)(G + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    HIP_TRY(gpsat::launch_glue(G, ndim, nvars, R, static_cast<const long long*>(h->sel_cnt.p), dp, dx, dv, sigma, sigma_rows ? dsig : nullptr, dout, h->stream));
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)nvars * G * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float km = 0.f;
    HIP_TRY(hipEventElapsedTime(&km, h->ev[1], h->ev[2]));
    h->last_kernel_ms = km; h->last_total_ms = km;
    return GPSAT_OK;
}

#ifdef GPSAT_PROFILE
// diagnostic build only: per-wave, per-segment cycle counters of the last call ([4 waves][16 slots])
int gpsat_debug_profile(gpsat_handle* h, unsigned long long* out64) {
    if (!h || !out64) return GPSAT_EINVAL;
    std::memcpy(out64, h->prof_host, 64 * sizeof(unsigned long long));
    return GPSAT_OK;
}
// per-workgroup first-tile start and kernel-exit times in 100 MHz ticks: [1024] starts then [1024] ends
int gpsat_debug_spans(gpsat_handle* h, unsigned long long* out2048) {
    if (!h || !out2048) return GPSAT_EINVAL;
    std::memcpy(out2048, h->prof_host + 64 + 8 * 1024, 2048 * sizeof(unsigned long long));
    return GPSAT_OK;
}
// event trace of the first evaluation of workgroup 0: [8 waves][1024] entries (cycle << 16 | arg << 8 | code), 0 = unused
int gpsat_debug_trace(gpsat_handle* h, unsigned long long* out8192) {
    if (!h || !out8192) return GPSAT_EINVAL;
    std::memcpy(out8192, h->prof_host + 64, 8 * 1024 * sizeof(unsigned long long));
    return GPSAT_OK;
}
#endif

}  // extern "C"
