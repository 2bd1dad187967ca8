/*
 * ORACLE -- test infrastructure only (see tron_dense.h header).
 *
 * Dense TRON for n <= TRON_NMAX.  All matrices are full symmetric, column
 * major with leading dimension n (as the reference's device code fills both
 * triangles: src/models/acopf/acopf_eval_linelimit_kernel_gpu.jl:573-574).
 * Sums run sequentially in index order; no FMA contraction (see Makefile).
 */
#include "tron_dense.h"
#include <math.h>
#include <string.h>

#define NM TRON_NMAX

void tron_default_options(tron_options *o)
{
    /* acopf_tron_linelimit_kernel.jl:44-51 and the call at
     * acopf_auglag_linelimit_kernel_gpu.jl:96 (500, 200, 1e-6);
     * CPU overrides: acopf_auglag_linelimit_kernel_cpu.jl:104-106 */
    o->max_feval = 500;
    o->max_minor = 200;
    o->gtol = 1e-6;
    o->frtol = 1e-12;
    o->fatol = 0.0;
    o->fmin = -1e32;
    o->cgtol = 0.1;
}

/* ---- BLAS-1 minis ------------------------------------------------------- */
static double ddot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

double tron_dnrm2(int n, const double *x)
{
    return sqrt(ddot(n, x, x));
}

static void daxpy(int n, double a, const double *x, double *y)
{
    for (int i = 0; i < n; i++) y[i] += a * x[i];
}

/* w = A x, A full symmetric */
static void dssyax(int n, const double *A, const double *x, double *w)
{
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += A[i + n * j] * x[j];
        w[i] = s;
    }
}

static void dmid(int n, double *x, const double *xl, const double *xu)
{
    for (int i = 0; i < n; i++) x[i] = fmax(xl[i], fmin(x[i], xu[i]));
}

/* ---- projected-gradient helpers ------------------------------------------ */
double tron_dgpnorm(int n, const double *x, const double *xl, const double *xu, const double *g)
{
    double r = 0.0;
    for (int i = 0; i < n; i++) {
        if (xl[i] != xu[i]) {
            if (x[i] == xl[i])
                r = fmax(r, fabs(fmin(g[i], 0.0)));
            else if (x[i] == xu[i])
                r = fmax(r, fabs(fmax(g[i], 0.0)));
            else
                r = fmax(r, fabs(g[i]));
        }
    }
    return r;
}

void tron_dbreakpt(int n, const double *x, const double *xl, const double *xu, const double *w,
                   int *nbrpt, double *brptmin, double *brptmax)
{
    int nb = 0;
    double bmin = 0.0, bmax = 0.0;
    for (int i = 0; i < n; i++) {
        double brpt;
        if (x[i] < xu[i] && w[i] > 0.0)
            brpt = (xu[i] - x[i]) / w[i];
        else if (x[i] > xl[i] && w[i] < 0.0)
            brpt = (xl[i] - x[i]) / w[i];
        else
            continue;
        nb++;
        if (nb == 1) {
            bmin = brpt;
            bmax = brpt;
        } else {
            bmin = fmin(brpt, bmin);
            bmax = fmax(brpt, bmax);
        }
    }
    if (nb == 0) {
        bmin = 0.0;
        bmax = 0.0;
    }
    *nbrpt = nb;
    *brptmin = bmin;
    *brptmax = bmax;
}

void tron_dgpstep(int n, const double *x, const double *xl, const double *xu, double alpha,
                  const double *w, double *s)
{
    for (int i = 0; i < n; i++) {
        double t = x[i] + alpha * w[i];
        if (t < xl[i])
            s[i] = xl[i] - x[i];
        else if (t > xu[i])
            s[i] = xu[i] - x[i];
        else
            s[i] = alpha * w[i];
    }
}

double tron_dtrqsol(int n, const double *x, const double *p, double delta)
{
    double ptx = ddot(n, p, x);
    double ptp = ddot(n, p, p);
    double xtx = ddot(n, x, x);
    double dsq = delta * delta;
    double rad = ptx * ptx + ptp * (dsq - xtx);
    rad = sqrt(fmax(rad, 0.0));
    if (ptx > 0.0) return (dsq - xtx) / (ptx + rad);
    if (rad > 0.0) return (rad - ptx) / ptp;
    return 0.0;
}

/* ---- Cholesky with diagonal scaling and shift (dicfs/dicf, dense) --------- */
/* in-place lower Cholesky of the lower triangle of L; returns 0 or -1 */
static int dicf(int n, double *L)
{
    for (int j = 0; j < n; j++) {
        for (int k = 0; k < j; k++) {
            double ljk = L[j + n * k];
            for (int i = j; i < n; i++) L[i + n * j] -= L[i + n * k] * ljk;
        }
        double d = L[j + n * j];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        L[j + n * j] = d;
        for (int i = j + 1; i < n; i++) L[i + n * j] /= d;
    }
    return 0;
}

int tron_dicfs(int n, double alpha, const double *A, double *L)
{
    const int nbmax = 3;
    const double alpham = 1.0e-3;
    const double nbfactor = 512.0;
    double wa1[NM], wa2[NM];

    /* l2 norms of the columns of A */
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += A[i + n * j] * A[i + n * j];
        wa1[j] = sqrt(s);
    }
    for (int i = 0; i < n; i++) wa2[i] = (wa1[i] > 0.0) ? 1.0 / sqrt(wa1[i]) : 1.0;

    double alphas = (alpha <= 0.0) ? alpham : alpha;

    /* initial shift */
    alpha = 0.0;
    for (int i = 0; i < n; i++) {
        double aii = A[i + n * i];
        if (aii == 0.0)
            alpha = alphas;
        else
            alpha = fmax(alpha, -aii * (wa2[i] * wa2[i]));
    }
    if (alpha > 0.0) alpha = fmax(alpha, alphas);

    int nb = 1;
    for (;;) {
        for (int j = 0; j < n; j++) {
            for (int i = 0; i < n; i++) L[i + n * j] = A[i + n * j] * wa2[i] * wa2[j];
            L[j + n * j] = A[j + n * j] * (wa2[j] * wa2[j]) + alpha;
        }
        int info = dicf(n, L);
        if (info >= 0) {
            if (alpha == alphas && nb < nbmax) {
                alphas = alphas / nbfactor;
                alpha = alphas;
                nb++;
            } else {
                for (int j = 0; j < n; j++)
                    for (int i = j; i < n; i++) L[i + n * j] = L[i + n * j] / wa2[i];
                /* zero the strict upper triangle for tidiness */
                for (int j = 0; j < n; j++)
                    for (int i = 0; i < j; i++) L[i + n * j] = 0.0;
                return nb;
            }
        } else {
            alpha = fmax(2.0 * alpha, alphas);
        }
    }
}

/* r <- L^{-1} r */
static void dstrsol_n(int n, const double *L, double *r)
{
    for (int j = 0; j < n; j++) {
        double t = r[j] / L[j + n * j];
        for (int k = j + 1; k < n; k++) r[k] -= L[k + n * j] * t;
        r[j] = t;
    }
}

/* r <- L^{-T} r */
static void dstrsol_t(int n, const double *L, double *r)
{
    r[n - 1] = r[n - 1] / L[(n - 1) + n * (n - 1)];
    for (int j = n - 2; j >= 0; j--) {
        double t = 0.0;
        for (int k = j + 1; k < n; k++) t += L[k + n * j] * r[k];
        r[j] = (r[j] - t) / L[j + n * j];
    }
}

/* ---- Cauchy step ---------------------------------------------------------- */
void tron_dcauchy(int n, const double *x, const double *xl, const double *xu, const double *A,
                  const double *g, double delta, double *alpha_io, double *s)
{
    const double mu0 = 0.01, interpf = 0.1, extrapf = 10.0;
    double wa[NM] = {0};
    double alpha = *alpha_io;
    int nbrpt;
    double brptmin, brptmax;

    for (int i = 0; i < n; i++) wa[i] = -g[i];
    tron_dbreakpt(n, x, xl, xu, wa, &nbrpt, &brptmin, &brptmax);

    int interp;
    tron_dgpstep(n, x, xl, xu, -alpha, g, s);
    if (tron_dnrm2(n, s) > delta) {
        interp = 1;
    } else {
        dssyax(n, A, s, wa);
        double gts = ddot(n, g, s);
        double q = 0.5 * ddot(n, s, wa) + gts;
        interp = (q >= mu0 * gts);
    }

    if (interp) {
        int search = 1;
        while (search) {
            alpha = interpf * alpha;
            tron_dgpstep(n, x, xl, xu, -alpha, g, s);
            if (tron_dnrm2(n, s) <= delta) {
                dssyax(n, A, s, wa);
                double gts = ddot(n, g, s);
                double q = 0.5 * ddot(n, s, wa) + gts;
                search = (q > mu0 * gts);
            }
        }
    } else {
        /* extrapolate while the step stays inside the region; keep the last
         * alpha that gave sufficient decrease ("recover the last successful step") */
        int search = 1;
        double alphas = alpha;
        while (search && alpha <= brptmax) {
            alpha = extrapf * alpha;
            tron_dgpstep(n, x, xl, xu, -alpha, g, s);
            if (tron_dnrm2(n, s) <= delta) {
                dssyax(n, A, s, wa);
                double gts = ddot(n, g, s);
                double q = 0.5 * ddot(n, s, wa) + gts;
                if (q < mu0 * gts) {
                    search = 1;
                    alphas = alpha;
                }
            } else {
                search = 0;
            }
        }
        alpha = alphas;
        tron_dgpstep(n, x, xl, xu, -alpha, g, s);
    }
    *alpha_io = alpha;
}

/* ---- preconditioned CG inside the trust region ----------------------------- */
int tron_dtrpcg(int n, const double *A, const double *g, double delta, const double *L,
                double tol, double stol, int itermax, double *w, int *iters_out)
{
    double p[NM], q[NM], r[NM], t[NM], z[NM];

    for (int i = 0; i < n; i++) w[i] = 0.0;
    for (int i = 0; i < n; i++) t[i] = -g[i];
    memcpy(r, t, sizeof(double) * n);
    dstrsol_n(n, L, r);
    memcpy(p, r, sizeof(double) * n);

    double rho = ddot(n, r, r);
    double rnorm0 = sqrt(rho);
    if (rnorm0 == 0.0) {
        *iters_out = 0;
        return 1;
    }

    for (int iters = 1; iters <= itermax; iters++) {
        memcpy(z, p, sizeof(double) * n);
        dstrsol_t(n, L, z);
        dssyax(n, A, z, q);
        memcpy(z, q, sizeof(double) * n);
        dstrsol_n(n, L, q);

        double ptq = ddot(n, p, q);
        double alpha = (ptq > 0.0) ? rho / ptq : 0.0;
        double sigma = tron_dtrqsol(n, w, p, delta);

        if (ptq <= 0.0 || alpha >= sigma) {
            daxpy(n, sigma, p, w);
            *iters_out = iters;
            return (ptq <= 0.0) ? 3 : 4;
        }

        daxpy(n, alpha, p, w);
        daxpy(n, -alpha, q, r);
        daxpy(n, -alpha, z, t);

        double rtr = ddot(n, r, r);
        double rnorm = sqrt(rtr);
        double tnorm = tron_dnrm2(n, t);
        if (tnorm <= tol) {
            *iters_out = iters;
            return 1;
        }
        if (rnorm <= stol) {
            *iters_out = iters;
            return 2;
        }
        double beta = rtr / rho;
        for (int i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
        rho = rtr;
    }
    *iters_out = itermax;
    return 5;
}

/* ---- projected search ------------------------------------------------------ */
void tron_dprsrch(int n, double *x, const double *xl, const double *xu, const double *A,
                  const double *g, double *w)
{
    const double mu0 = 0.01, interpf = 0.5;
    double wa1[NM], wa2[NM];
    double alpha = 1.0;
    int nbrpt;
    double brptmin, brptmax;

    tron_dbreakpt(n, x, xl, xu, w, &nbrpt, &brptmin, &brptmax);

    int search = 1;
    while (search && alpha > brptmin) {
        tron_dgpstep(n, x, xl, xu, alpha, w, wa1);
        dssyax(n, A, wa1, wa2);
        double gts = ddot(n, g, wa1);
        double q = 0.5 * ddot(n, wa1, wa2) + gts;
        if (q <= mu0 * gts)
            search = 0;
        else
            alpha = interpf * alpha;
    }
    if (alpha < 1.0 && alpha < brptmin) alpha = brptmin;

    tron_dgpstep(n, x, xl, xu, alpha, w, wa1);
    daxpy(n, alpha, w, x);
    dmid(n, x, xl, xu);
    memcpy(w, wa1, sizeof(double) * n);
}

/* ---- subspace projected CG -------------------------------------------------- */
int tron_dspcg(int n, double *x, const double *xl, const double *xu, const double *A,
               const double *g, double delta, double rtol, double *s, int itermax, int *iters_out)
{
    double w[NM], gfree[NM], wa[NM], xf[NM], xlf[NM], xuf[NM];
    double B[NM * NM], L[NM * NM];
    int indfree[NM];
    int iters = 0;

    dssyax(n, A, s, w);
    daxpy(n, 1.0, s, x);
    dmid(n, x, xl, xu);

    for (int nfaces = 1; nfaces <= n; nfaces++) {
        int nfree = 0;
        for (int i = 0; i < n; i++)
            if (xl[i] < x[i] && x[i] < xu[i]) indfree[nfree++] = i;
        if (nfree == 0) {
            *iters_out = iters;
            return 1;
        }

        for (int j = 0; j < nfree; j++)
            for (int i = 0; i < nfree; i++) B[i + nfree * j] = A[indfree[i] + n * indfree[j]];

        tron_dicfs(nfree, 0.0, B, L);

        for (int j = 0; j < nfree; j++) {
            gfree[j] = w[indfree[j]] + g[indfree[j]];
            wa[j] = g[indfree[j]];
        }
        double gfnorm = tron_dnrm2(nfree, wa);

        double tol = rtol * gfnorm;
        int itertr;
        int infotr = tron_dtrpcg(nfree, B, gfree, delta, L, tol, 0.0, itermax, w, &itertr);
        iters += itertr;
        dstrsol_t(nfree, L, w);

        for (int j = 0; j < nfree; j++) {
            xf[j] = x[indfree[j]];
            xlf[j] = xl[indfree[j]];
            xuf[j] = xu[indfree[j]];
        }
        tron_dprsrch(nfree, xf, xlf, xuf, B, gfree, w);

        for (int j = 0; j < nfree; j++) {
            x[indfree[j]] = xf[j];
            s[indfree[j]] += w[j];
        }

        dssyax(n, A, s, w);
        for (int j = 0; j < nfree; j++) gfree[j] = w[indfree[j]] + g[indfree[j]];
        double gfnormf = tron_dnrm2(nfree, gfree);

        if (gfnormf <= rtol * gfnorm) {
            *iters_out = iters;
            return 1;
        } else if (infotr == 3 || infotr == 4) {
            *iters_out = iters;
            return 2;
        } else if (iters > itermax) {
            *iters_out = iters;
            return 3;
        }
    }
    *iters_out = iters;
    return 0;
}

/* ---- trust-region Newton state machine --------------------------------------- */
double tron_dtron(int n, double *x, const double *xl, const double *xu, double f, const double *g,
                  const double *A, double frtol, double fatol, double fmin_, double cgtol,
                  int itermax, double delta, int *task_io, tron_state *st)
{
    const double eta0 = 1e-4, eta1 = 0.25, eta2 = 0.75;
    const double sigma1 = 0.25, sigma2 = 0.5, sigma3 = 4.0;
    int task = *task_io;
    int work, iter, iterscg;
    double fc = 0.0, alphac, prered = 0.0;

    if (task == TRON_START) {
        iter = 1;
        iterscg = 0;
        alphac = 1.0;
        work = 1;
    } else {
        work = st->work;
        iter = st->iter;
        iterscg = st->iterscg;
        fc = st->fc;
        alphac = st->alphac;
        prered = st->prered;
    }

    int search = 1;
    while (search) {
        if (work == 1) { /* COMPUTE */
            double wa[NM];
            int iters;
            fc = f;
            memcpy(st->xc, x, sizeof(double) * n);
            tron_dcauchy(n, x, xl, xu, A, g, delta, &alphac, st->s);
            tron_dspcg(n, x, xl, xu, A, g, delta, cgtol, st->s, itermax, &iters);
            dssyax(n, A, st->s, wa);
            prered = -(ddot(n, st->s, g) + 0.5 * ddot(n, st->s, wa));
            iterscg += iters;
            task = TRON_F;
        }

        if (work == 2) { /* EVALUATE */
            double actred = fc - f;
            double snorm = tron_dnrm2(n, st->s);
            if (iter == 1) delta = fmin(delta, snorm);

            double g0 = ddot(n, g, st->s);
            double alpha;
            if (f - fc - g0 <= 0.0)
                alpha = sigma3;
            else
                alpha = fmax(sigma1, -0.5 * (g0 / (f - fc - g0)));

            if (actred < eta0 * prered)
                delta = fmin(fmax(alpha, sigma1) * snorm, sigma2 * delta);
            else if (actred < eta1 * prered)
                delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma2 * delta));
            else if (actred < eta2 * prered)
                delta = fmax(sigma1 * delta, fmin(alpha * snorm, sigma3 * delta));
            else
                delta = fmax(delta, fmin(alpha * snorm, sigma3 * delta));

            if (actred > eta0 * prered) {
                task = TRON_GH;
                iter++;
            } else {
                task = TRON_F;
                memcpy(x, st->xc, sizeof(double) * n);
                f = fc;
            }

            if (f < fmin_) task = TRON_WARN;
            if (fabs(actred) <= fatol && prered <= fatol) task = TRON_CONV;
            if (fabs(actred) <= frtol * fabs(f) && prered <= frtol * fabs(f)) task = TRON_CONV;
        }

        if (task == TRON_F && work == 2) {
            search = 1;
            work = 1;
        } else {
            search = 0;
        }
    }

    if (work == 3) task = TRON_NEWX;

    if (task == TRON_F && work == 1)
        work = 2;
    else if (task == TRON_F && work == 2)
        work = 1;
    else if (task == TRON_GH)
        work = 3;
    else if (task == TRON_NEWX)
        work = 1;

    st->work = work;
    st->iter = iter;
    st->iterscg = iterscg;
    st->fc = fc;
    st->alphac = alphac;
    st->prered = prered;
    *task_io = task;
    return delta;
}

/* optional debugging hook (NULL by default) */
void (*tron_trace)(int n, int minor_iter, int nfev, int task, double f, double delta,
                   const double *x, const double *g) = 0;

/* ---- reverse-communication driver ---------------------------------------------
 * mirrors acopf_tron_linelimit_kernel.jl:61-144 */
int tron_solve(int n, double *x, const double *xl, const double *xu,
               tron_eval_f ef, tron_eval_gh egh, void *ctx,
               const tron_options *opt, int *minor_iter_out, int *nfev_out)
{
    double g[NM], A[NM * NM];
    tron_state st;
    memset(&st, 0, sizeof(st));
    memset(A, 0, sizeof(A));
    memset(g, 0, sizeof(g));

    int task = TRON_START, status = 0;
    double delta = 0.0, f = 0.0;
    int nfev = 0, minor_iter = 0;
    int search = 1;
    const int cg_itermax = n;

    while (search) {
        if (task == TRON_START || task == TRON_F) {
            f = ef(x, ctx);
            nfev++;
            if (nfev >= opt->max_feval) search = 0;
        }
        if (task == TRON_START || task == TRON_GH) {
            egh(x, g, A, ctx);
            minor_iter++;
        }
        if (task == TRON_START) delta = tron_dnrm2(n, g);

        if (search)
            delta = tron_dtron(n, x, xl, xu, f, g, A, opt->frtol, opt->fatol, opt->fmin, opt->cgtol,
                               cg_itermax, delta, &task, &st);

        if (tron_trace) tron_trace(n, minor_iter, nfev, task, f, delta, x, g);
        if (task == TRON_NEWX) {
            double gnorm_inf = tron_dgpnorm(n, x, xl, xu, g);
            if (gnorm_inf <= opt->gtol) task = TRON_CONV;
            if (minor_iter >= opt->max_minor) {
                status = 1;
                search = 0;
            }
        }
        if (task == TRON_CONV || task == TRON_WARN) search = 0;
    }
    *minor_iter_out = minor_iter;
    if (nfev_out) *nfev_out = nfev;
    return status;
}
