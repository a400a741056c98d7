/*
 * gpsat_hip.h -- C ABI of the MI355X-native local-expert exact-GP engine (libgpsat_hip.so).
 *
 * Drop-in boundary.  The reference (CPOMUCL/GPSat, pure Python) has no FFI; the backend it
 * calls per expert tile is a Python class satisfying BaseGPRModel
 * (GPSat/models/base_model.py:17-82) selected through model_config["oi_model"]
 * (GPSat/models/__init__.py:3-28, GPSat/local_experts.py:292-346).  Each entry point below
 * states which reference interface it replaces; the Python binding a maintainer would add
 * on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; all pointers are caller-owned;
 *  - every function returns 0 on success or a negative GPSAT_E* code and never throws;
 *    gpsat_last_error() returns a thread-local message for the last failure;
 *  - one handle per GPU; calls on one handle must be serialised by the caller, distinct
 *    handles may be driven from distinct threads / processes;
 *  - calls are host-synchronous on return.
 *
 * Parameter vector order (H = D + 2):
 *     theta = (lengthscale_0 .. lengthscale_{D-1}, kernel_variance, likelihood_variance)
 * i.e. the reference's param_names ["lengthscales","kernel_variance","likelihood_variance"]
 * (GPSat/models/gpflow_models.py:179-184) flattened.
 */
#ifndef GPSAT_HIP_H
#define GPSAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPSAT_ABI_VERSION 3

/* error codes */
#define GPSAT_OK            0
#define GPSAT_EINVAL       -1   /* bad argument / unsupported configuration */
#define GPSAT_ENODEV       -2   /* no usable HIP device                      */
#define GPSAT_ENOMEM       -3   /* device allocation failed                   */
#define GPSAT_EHIP         -4   /* HIP runtime error (see gpsat_last_error)   */

/* compute dtype of the bulk arrays (X, y, Xs, f_mean, f_var, y_var) */
#define GPSAT_F32 0
#define GPSAT_F64 1

/* kernels: gpflow.kernels names accepted by GPflowGPRModel (gpflow_models.py:72-75,116-135) */
#define GPSAT_KERNEL_RBF      0   /* "RBF" / "SquaredExponential" */
#define GPSAT_KERNEL_MATERN12 1   /* "Matern12" / "Exponential"   */
#define GPSAT_KERNEL_MATERN32 2   /* "Matern32" (reference default, gpflow_models.py:44) */
#define GPSAT_KERNEL_MATERN52 3   /* "Matern52" */

/* optimisers */
#define GPSAT_OPT_NONE  0   /* optimise=False in LocalExpertOI.run (local_experts.py:1126-1132) */
#define GPSAT_OPT_LBFGS 1   /* replaces gpflow.optimizers.Scipy / L-BFGS-B (gpflow_models.py:317-321) */
#define GPSAT_OPT_ADAM  2   /* fixed-step alternative named by BASELINE.json north_star */

/* where the bulk arrays live */
#define GPSAT_MEM_HOST   0
#define GPSAT_MEM_DEVICE 1

/* per-tile status */
#define GPSAT_STATUS_CONVERGED 0   /* optimiser met ftol / gtol (scipy success=True); also: a line */
                                   /*   search failed after a step whose decrease was already at   */
                                   /*   the resolution of the arithmetic (1e-3 |f| in fp32, 1e-12   */
                                   /*   |f| in fp64) -- not with ftol switched off                  */
#define GPSAT_STATUS_MAXITER   1   /* iteration limit reached (scipy success=False)           */
#define GPSAT_STATUS_NOT_PD    2   /* Cholesky failed at the initial / final parameters        */
#define GPSAT_STATUS_NAN       3   /* NaN encountered                                          */
#define GPSAT_STATUS_SKIPPED   4   /* tile had no observations                                 */
#define GPSAT_STATUS_NOT_OPTIMISED 5 /* optimiser == NONE: objective + predict only            */
#define GPSAT_STATUS_LS_FAILED 6   /* line search failed with an empty L-BFGS history (scipy     */
                                   /*   ABNORMAL_TERMINATION_IN_LNSRCH, success=False); theta is */
                                   /*   the best sufficient-decrease point seen                   */

typedef struct gpsat_handle gpsat_handle;

typedef struct gpsat_opts {
    int32_t workgroups_per_cu;   /* persistent workgroups per CU (0 = default 2)            */
    int32_t reserved[7];
} gpsat_opts;

/*
 * One packed ragged batch of T independent expert tiles.
 *
 * Replaces, for all T tiles at once, the per-tile sequence of LocalExpertOI.run
 * (GPSat/local_experts.py:1043-1159):
 *   model = Model(data=df_local, ...)            -> X, y (already scaled / de-meaned by the host
 *                                                   exactly as base_model.py:243-245, in fp64,
 *                                                   then cast to `dtype`)
 *   model.set_parameters / load_params            -> theta0
 *   model.set_parameter_constraints(...)          -> lo, hi (already divided by coords_scale and
 *                                                   with theta0 moved within tol on the host,
 *                                                   gpflow_models.py:459-479)
 *   model.optimise_parameters(**optim_kwargs)     -> optimiser, max_iter, trainable
 *   model.get_objective_function_value()          -> nll
 *   model.get_parameters()                        -> theta
 *   model.predict(coords=prediction_coords)       -> f_mean ("f*"), f_var ("f*_var"), y_var
 */
typedef struct gpsat_batch {
    /* ---- shape ---- */
    int32_t T;                 /* number of tiles                                           */
    int32_t D;                 /* input dimension (1..4 in this build)                      */
    int32_t dtype;             /* GPSAT_F32 (fp32 MFMA kernels) | GPSAT_F64 (fp64 MFMA kernels) */
    int32_t kernel;            /* GPSAT_KERNEL_*                                            */
    int32_t memory;            /* GPSAT_MEM_HOST / GPSAT_MEM_DEVICE for the bulk arrays      */
    int32_t optimiser;         /* GPSAT_OPT_*                                               */
    int32_t max_iter;          /* optimiser iteration limit (scipy options.maxiter)         */
    int32_t max_ls;            /* max line-search evaluations per iteration (0 = 20, scipy maxls) */
    double  ftol;              /* relative objective decrease tolerance (0 = default: fp64   */
                               /*   2.2e-9 = SciPy's factr*eps, fp32 1e-6 = its fp32 analogue; */
                               /*   negative = never stop on this criterion)                  */
    double  gtol;              /* max-norm gradient tolerance in u-space (0 = default 1e-5,  */
                               /*   scipy pgtol; negative = never stop on this criterion)     */
    double  adam_lr;           /* Adam learning rate (0 = default 0.1)                      */

    /* ---- metadata: ALWAYS host memory ---- */
    const int64_t *obs_off;    /* [T+1] CSR offsets into X / y        (rows)                */
    const int64_t *pred_off;   /* [T+1] CSR offsets into Xs / outputs (rows)                */
    const double  *theta0;     /* [T*H] initial parameters, constrained space                */
    const double  *lo;         /* [T*H] lower bounds; NaN/inf => unconstrained (softplus)    */
    const double  *hi;         /* [T*H] upper bounds                                          */
    const uint8_t *trainable;  /* [H]   0 => parameter fixed (optim_kwargs.fixed_params)      */

    /* ---- bulk inputs: host or device according to `memory`, element type `dtype` ---- */
    const void *X;             /* [sum N, D] row-major, scaled coordinates                   */
    const void *y;             /* [sum N]    de-meaned / scaled observations                 */
    const void *Xs;            /* [sum P, D] row-major, scaled prediction coordinates        */

    /* ---- outputs ---- */
    double  *theta;            /* [T*H] host: learned parameters                             */
    double  *nll;              /* [T]   host: objective (negative log marginal likelihood)   */
    double  *grad;             /* [T*H] host, optional (may be NULL): dNLL/dtheta at `theta` */
    int32_t *status;           /* [T]   host: GPSAT_STATUS_*                                 */
    int32_t *n_eval;           /* [T]   host: objective+gradient evaluations performed       */
    void    *f_mean;           /* [sum P] host|device (as `memory`): "f*"                    */
    void    *f_var;            /* [sum P] "f*_var"                                           */
    void    *y_var;            /* [sum P] "y_var"                                            */

    /* ---- optional full posterior covariance (predict(full_cov=True), gpflow_models.py:245-263); ABI >= 2 ---- */
    const int64_t *cov_off;    /* [T+1] host: element offsets into f_cov, cov_off[t+1]-cov_off[t] = P_t^2; NULL = off */
    void    *f_cov;            /* [sum P_t^2] host|device (as `memory`), element type `dtype`: per tile the      */
                               /*   row-major P_t x P_t matrix "f*_cov" = K** - K*^T K^-1 K*; NULL = not wanted   */

    /* ---- ABI >= 3 ---- */
    int32_t *n_iter;           /* [T] host, optional (may be NULL): optimiser iterations completed (scipy nit)  */
} gpsat_batch;

/* library / ABI version (GPSAT_ABI_VERSION) */
int gpsat_version(void);

/* thread-local description of the last error returned on this thread */
const char *gpsat_last_error(void);

/* number of HIP devices visible (0 when there is none); never fails */
int gpsat_device_count(void);

/*
 * Largest number of observations one tile may hold for `dtype` (GPSAT_F32 / GPSAT_F64) and input dimension D: the
 * tile's coordinates, observations and solve vectors stay in the 160 KiB LDS of its CU for the whole fit (D = 3:
 * 2,800 in fp64 -- the reference's published N = 2,500 fp64 fit, docs/notebooks/using_gpus.ipynb:77,165, fits -- and
 * 3,168 in fp32).  A batch holding a larger tile is refused with GPSAT_EINVAL.  0 for unsupported arguments.  ABI >= 3.
 */
int gpsat_max_tile_obs(int dtype, int D);

/*
 * Create an engine bound to one GPU.  Replaces model construction-time device discovery
 * (BaseGPRModel._get_device_names, base_model.py:279-300).  `opts` may be NULL.
 */
int gpsat_create(int device_id, const gpsat_opts *opts, gpsat_handle **out);

/* device name of the handle's GPU (reference: model.gpu_name, local_experts.py:1180) */
int gpsat_device_name(gpsat_handle *h, char *buf, int buflen);

/* release every device resource owned by the handle */
int gpsat_destroy(gpsat_handle *h);

/* fit (optional) + objective + predict for one packed batch; see gpsat_batch */
int gpsat_fit_predict_batch(gpsat_handle *h, const gpsat_batch *b);

/*
 * Batched tile selection: which rows of a point table belong to each of T expert tiles.
 * Replaces DataLoader.local_data_select (GPSat/dataloader.py:2352-2447) and the max_dist filter of
 * PredictionLocations (GPSat/prediction_locations.py:18-43) for all experts at once, in fp64 with the reference's
 * comparison arithmetic (bit-exact membership, source row order).
 *   kind 0: points[cols[k][0]] <comp> (refs[cols[k][0]] + val[k])       comp: 0 >=, 1 >, 2 ==, 3 <, 4 <=
 *   kind 1: sum_m (points[cols[k][m]] - refs[cols[k][m]])^2 <= val[k]^2  (comp 4, observations: inclusive ball)
 *                                                          <  val[k]^2  (comp 3, prediction locations: strict)
 */
#define GPSAT_SEL_MAXCRIT 4
typedef struct gpsat_select_spec {
    int32_t n_crit;                        /* 1 .. GPSAT_SEL_MAXCRIT criteria, ANDed                 */
    int32_t kind[GPSAT_SEL_MAXCRIT];
    int32_t comp[GPSAT_SEL_MAXCRIT];
    int32_t ncols[GPSAT_SEL_MAXCRIT];      /* kind 1: 1..3 columns                                   */
    int32_t cols[GPSAT_SEL_MAXCRIT][3];    /* column indices (same numbering in points and refs)      */
    double  val[GPSAT_SEL_MAXCRIT];
} gpsat_select_spec;

/*
 * points: host, column-major [C][M] fp64;  refs: host, row-major [T][C] fp64.
 * off  : host out [T+1] CSR offsets (always written).
 * idx  : host out [capacity] selected row indices per expert, ascending; may be NULL (count only).
 * Returns GPSAT_EINVAL if capacity < off[T] (off is still valid: call again with a larger buffer).
 * Two-call use (sizes with idx = NULL, then the same arguments with idx): the indices of the first call stay on the device
 * and the second call only copies them out -- PROVIDED `points` and `refs` are the same buffers with unchanged contents and
 * no other call was made on the handle in between.  The library checks the arguments and a fingerprint of the contents (all
 * of refs, a sample of points) and selects again when anything differs; do not rely on the sample to catch a partial refill
 * of `points`.
 */
int gpsat_select_batch(gpsat_handle *h, const gpsat_select_spec *spec, int64_t M, int32_t C, const double *points,
                       int32_t T, const double *refs, int64_t *off, int32_t *idx, int64_t capacity);

/*
 * Gaussian smoothing of one hyper-parameter field over T expert locations (one "other dimensions" slice).
 * Replaces gaussian_2d_weight (GPSat/postprocessing.py:22-52) as called by smooth_hyperparameters (:277-288, after
 * the min/max clipping done by the caller).  x, y, vals, out: host fp64 [T]; NaN vals are skipped; out is NaN when
 * all weights vanish.
 */
int gpsat_smooth_batch(gpsat_handle *h, int32_t T, const double *x, const double *y, const double *vals, double l_x,
                       double l_y, double *out);

/*
 * Gluing of overlapping local predictions (GPSat/postprocessing.py:447-577): R prediction rows pre-sorted by
 * prediction location into G segments seg[G+1]; pred, xprt: host fp64 [ndim][R] (ndim 1 or 2); vals: host fp64
 * [nvars][R] (nvars <= 4); sigma = inference_radius / R_factor, or per row in sigma_rows [R] when not NULL (the
 * per-expert inference_radius dict of :490-493); out: host fp64 [nvars][G] = sum w v / sum w with
 * w = prod_d normpdf(pred_d; xprt_d, sigma).
 */
int gpsat_glue_batch(gpsat_handle *h, int64_t R, int32_t G, int32_t ndim, int32_t nvars, const int64_t *seg,
                     const double *pred, const double *xprt, const double *vals, double sigma,
                     const double *sigma_rows, double *out);

/*
 * Timing of the last gpsat_fit_predict_batch on this handle, measured with HIP events on the
 * handle's stream: kernel_ms = the persistent tile kernel alone, total_ms = H2D + kernel + D2H.
 */
int gpsat_last_timing(gpsat_handle *h, double *kernel_ms, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* GPSAT_HIP_H */
