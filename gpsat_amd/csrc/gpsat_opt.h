// gpsat_opt.h -- per-workgroup optimiser state and the on-device L-BFGS / Adam driver shared by the fp32 and fp64
// tile kernels (device code, thread 0 of a workgroup, fp64 arithmetic).  Included inside namespace gpsat.
#ifndef GPSAT_OPT_H
#define GPSAT_OPT_H

#ifndef GPSAT_NW
#define GPSAT_NW 4
#endif
constexpr int NW = GPSAT_NW;   // waves per workgroup (power of two; fp32 kernels 4, fp64 kernels see gpsat_kernels_f64.hip)
constexpr int NT = 64 * NW;    // threads per workgroup
constexpr int HMAX = 6;        // max D + 2 (D <= 4)
constexpr int MH = 10;         // L-BFGS history (SciPy L-BFGS-B maxcor default)

// ---------------------------------------------------------------------------------------------
// per-workgroup state
// ---------------------------------------------------------------------------------------------
struct Shared {
    // evaluation interface
    double theta[HMAX];
    double gth[HMAX];          // dNLL/dtheta
    double nll;
    double logdet;
    double red[8][8];          // partial sums of eight VIRTUAL waves (the 4-wave builds run two each): same sums in every build
    // optimiser state (thread 0 writes, everybody reads after a barrier)
    double lo[HMAX], hi[HMAX], shift[HMAX];
    double u[HMAX], g[HMAX], f;            // current accepted point (u-space)
    double ut[HMAX], gt[HMAX], ft;         // trial point
    double ub[HMAX], gb[HMAX];             // best sufficient-decrease point of the running line search (value f_best)
    double d[HMAX];
    double S[MH][HMAX], Y[MH][HMAX], rho_[MH];
    double m1[HMAX], m2[HMAX];             // Adam moments
    // line search
    double t, t_prev, f_prev, dphi_prev, t_lo, f_lo, dphi_lo, t_hi, f_hi, dphi_hi, dphi0, t_best, f_best, last_dec;
    int ls_phase, ls_iter, ls_done, ls_ok;
    int hist_n, hist_pos;
    int trainable[HMAX];
    int box[HMAX];
    int fail, done, status, n_eval, n_eval_opt, iter, phase, want_grad;
    int tile;
    // flags of the Cholesky / inverse sweep (phase_pt of the fp32 kernels; the fp64 kernels use g0done and gnext)
    int g0done;                 // panel index up to which group 0 of the previous panel is in memory
    int gnext[2];               // fp64 kernels: dynamic group queue heads of the PT slots (alternating)
#ifdef GPSAT_PT_MAXNB
    int ready, parked, whfree;  // panels: chain complete / group-0 k-loop parked / parked k-loop consumed
    int gdone[2];               // groups of the panel of that parity that are finished
    int qhead;                  // bulk group queue head (all panels)
    int colrow[GPSAT_PT_MAXNB]; // per block column: panels whose rows are in memory
#endif
    int gradnext;               // dynamic group queue head of the gradient phase
    int coop_seq;               // cooperative tiles: phases this workgroup has opened as an owner (not part of a tile's state)
    int coop_now;               // the running evaluation is cooperative
    int hp[8];                  // helper bookkeeping (see helper_loop of the fp32 kernels)
#ifdef GPSAT_PROFILE
    unsigned long long prof[NW * 16];      // diagnostic build: cycle counters per wave and code segment
    int tcnt[NW], tron;                    // event trace (workgroup 0): entries per wave, on/off
#else
    unsigned long long prof[1];
#endif
};


// ---------------------------------------------------------------------------------------------
// parameter transforms (reference GPSat/utils.py:2320-2400,
// GPSat/models/gpflow_models.py:490-494): box -> lo + (hi-lo) sigmoid(u); else softplus(u) + shift
// ---------------------------------------------------------------------------------------------
static __device__ inline double softplus_d(double x) { return log1p(exp(-fabs(x))) + fmax(x, 0.0); }

static __device__ __noinline__ double theta_of_u(const Shared* sh, int i, double u) {
    if (sh->box[i]) return sh->lo[i] + (sh->hi[i] - sh->lo[i]) / (1.0 + exp(-u));
    return softplus_d(u) + sh->shift[i];
}

static __device__ __noinline__ double u_of_theta(const Shared* sh, int i, double th) {
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

static __device__ inline double dtheta_du(const Shared* sh, int i, double th) {
    if (sh->box[i]) return (th - sh->lo[i]) * (sh->hi[i] - th) / (sh->hi[i] - sh->lo[i]);
    return -expm1(-(th - sh->shift[i]));
}

// thread 0: trial u -> theta for the next evaluation
static __device__ __noinline__ void set_trial(Shared* sh, int H, const double* u) {
    for (int i = 0; i < H; ++i) {
        sh->ut[i] = u[i];
        if (sh->trainable[i]) sh->theta[i] = theta_of_u(sh, i, u[i]);
    }
}

// thread 0: after an evaluation, chain the gradient to u-space at the trial point
static __device__ __noinline__ void fetch_trial(Shared* sh, int H) {
    sh->ft = sh->fail ? __builtin_inf() : sh->nll;
    for (int i = 0; i < H; ++i)
        sh->gt[i] = (sh->trainable[i] && !sh->fail) ? sh->gth[i] * dtheta_du(sh, i, sh->theta[i]) : 0.0;
}

// L-BFGS two-loop recursion (thread 0): d = -H g
static __device__ __noinline__ void lbfgs_direction(Shared* sh, int H) {
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

static __device__ inline double cubic_min(double a, double fa, double da, double b, double fb, double db) {
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
static __device__ __noinline__ void ls_step(Shared* sh, int H, int max_ls) {
    const double c1 = 1e-4, c2 = 0.9;
    const double t = sh->t, ft = sh->ft;
    double dphit = 0.0;
    for (int i = 0; i < H; ++i) dphit += sh->gt[i] * sh->d[i];
    const bool finite = (ft == ft) && (ft < 1e300);
    const bool armijo = finite && (ft <= sh->f + c1 * t * sh->dphi0);
    if (armijo && ft < sh->f_best) {
        sh->f_best = ft; sh->t_best = t;
        for (int i = 0; i < H; ++i) { sh->ub[i] = sh->ut[i]; sh->gb[i] = sh->gt[i]; }
    }
    sh->ls_iter += 1;
    if (armijo && fabs(dphit) <= -c2 * sh->dphi0) { sh->ls_done = 1; return; }
    if (sh->ls_phase == 1) {
        // MINPACK-2 dcsrch's "XTOL TEST SATISFIED" (xtol = 0.1 in L-BFGS-B): this trial was placed inside a bracket
        // whose relative width had already shrunk to 10 % -- the search ends here, and L-BFGS-B takes the step
        // (lnsrlb treats the warning like convergence).  The sufficient-decrease point is taken: this trial if it is
        // one, else the best one seen; with none at all the search has failed.
        const double a0 = fmin(sh->t_lo, sh->t_hi), b0 = fmax(sh->t_lo, sh->t_hi);
        if (b0 - a0 <= 0.1 * b0) {
            if (armijo) { sh->ls_done = 1; return; }
            if (sh->t_best > 0.0) {
                sh->t = sh->t_best; sh->ft = sh->f_best;
                for (int i = 0; i < H; ++i) { sh->ut[i] = sh->ub[i]; sh->gt[i] = sh->gb[i]; }
                sh->ls_done = 3;          // accepted, but the factorisation in memory belongs to another point
                return;
            }
            sh->ls_done = 2;
            return;
        }
    }
    // max_ls evaluations without meeting the strong Wolfe conditions: failure, as L-BFGS-B's `iback >= maxls` (mainlb)
    if (sh->ls_iter >= max_ls) { sh->ls_done = 2; return; }
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
            // not bracketed yet (sufficient decrease, still descending): extrapolate as More-Thuente / SciPy's dcsrch do --
            // the cubic through the last two points when its minimiser lies ahead, safeguarded to
            // [t + 1.1 (t - t_prev), t + 4 (t - t_prev)]
            const double tp = sh->t_prev, dt = t - tp;
            double tn = cubic_min(tp, sh->f_prev, sh->dphi_prev, t, ft, dphit);
            const double lo_b = t + 1.1 * dt, hi_b = t + 4.0 * dt;
            if (!(tn > lo_b)) tn = hi_b;          // minimiser behind us or undefined: the cubic has no minimum ahead
            tn = fmin(tn, hi_b);
            sh->t_prev = t; sh->f_prev = ft; sh->dphi_prev = dphit;
            sh->t = tn;
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
// per-tile status codes written by the optimiser (include/gpsat_hip.h GPSAT_STATUS_*)
enum { ST_CONVERGED = 0, ST_MAXITER = 1, ST_LS_FAILED = 6 };

enum { PH_INIT = 0, PH_LS = 1, PH_ADAM = 2, PH_FINAL = 3, PH_EXIT = 4 };

struct OptCfg { int optimiser, max_iter, max_ls, want_grad_out; double ftol, gtol, adam_lr, noise_rel; };

// the accepted point is sh->u; decide whether the factorisation in memory already belongs to it
static __device__ __noinline__ void opt_finish(Shared* sh, int H, const OptCfg& o, bool factor_is_current) {
    sh->n_eval_opt = sh->n_eval;
    if (factor_is_current && !sh->fail) { sh->phase = PH_EXIT; return; }
    set_trial(sh, H, sh->u);
    sh->want_grad = o.want_grad_out;
    sh->phase = PH_FINAL;
}

static __device__ __noinline__ void opt_start_iteration(Shared* sh, int H, const OptCfg& o) {
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

static __device__ __noinline__ void opt_advance(Shared* sh, int H, const OptCfg& o) {
    switch (sh->phase) {
        case PH_INIT: {
            set_trial(sh, H, sh->u);
            fetch_trial(sh, H);
            sh->f = sh->ft;
            for (int i = 0; i < H; ++i) sh->g[i] = sh->gt[i];
            if (sh->fail) { sh->status = 2; sh->n_eval_opt = sh->n_eval; sh->phase = PH_EXIT; return; }
            sh->status = 1;
            if (o.optimiser == 1 && o.gtol > 0.0) {
                // L-BFGS-B tests the gradient norm before the first iteration too
                double gmax0 = 0.0;
                for (int i = 0; i < H; ++i) gmax0 = fmax(gmax0, fabs(sh->g[i]));
                if (gmax0 <= o.gtol) { sh->status = 0; opt_finish(sh, H, o, true); return; }
            }
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
            if (sh->ls_done == 1 || sh->ls_done == 3) {
                // accept the trial point (ls_done 1: the last evaluated one; 3: an earlier one, restored into ut / gt / ft)
                const bool current = sh->ls_done == 1;
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
                sh->last_dec = fold - fnew;
                for (int i = 0; i < H; ++i) { sh->u[i] = sh->ut[i]; sh->g[i] = sh->gt[i]; gmax = fmax(gmax, fabs(sh->gt[i])); }
                sh->iter += 1;
                const double den = fmax(fmax(fabs(fold), fabs(fnew)), 1.0);
                if ((fold - fnew) <= o.ftol * den || gmax <= o.gtol) { sh->status = 0; opt_finish(sh, H, o, current); return; }
                if (sh->iter >= o.max_iter) { sh->status = 1; opt_finish(sh, H, o, current); return; }
                opt_start_iteration(sh, H, o);
                return;
            }
            // Line search failed (no step satisfying the strong Wolfe conditions within max_ls evaluations).
            //  * The last accepted step's decrease was already at the resolution of the arithmetic (noise_rel * |f|: fp32
            //    1e-3 -- the objective carries rounding ~ cond(K) eps N --, fp64 1e-12): a further decrease cannot be told
            //    from noise.  That is the finite-precision form of the ftol test: converged.  (Not with ftol switched off.)
            //  * Otherwise as L-BFGS-B (mainlb, info != 0): with a non-empty history, discard it and restart once from
            //    steepest descent at the accepted point (the restart is not an iteration); with an empty history give up
            //    -- SciPy reports ABNORMAL_TERMINATION_IN_LNSRCH, success=False, hence a status of its own.  The best
            //    sufficient-decrease point seen by the failed search (if any) is kept rather than thrown away.
            if (sh->iter > 0 && o.ftol >= 0.0 && sh->last_dec <= o.noise_rel * fmax(fabs(sh->f), 1.0)) {
                sh->status = ST_CONVERGED;
                opt_finish(sh, H, o, false);
                return;
            }
            if (sh->hist_n > 0) {
                sh->hist_n = 0;
                opt_start_iteration(sh, H, o);
                return;
            }
            sh->status = ST_LS_FAILED;
            if (sh->t_best > 0.0 && sh->f_best < sh->f) {
                sh->last_dec = sh->f - sh->f_best;
                sh->f = sh->f_best;
                for (int i = 0; i < H; ++i) { sh->u[i] = sh->ub[i]; sh->g[i] = sh->gb[i]; }
            }
            opt_finish(sh, H, o, false);
            return;
        }
        default:  // PH_FINAL
            sh->phase = PH_EXIT;
            return;
    }
}


#endif  // GPSAT_OPT_H
