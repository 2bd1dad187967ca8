/*
 * ORACLE -- test infrastructure only.  Nothing under oracle/ is linked into,
 * imported by, or shipped with the product library (exaadmm.jl_amd/).
 *
 * Dense bound-constrained trust-region Newton method (TRON, Lin & More',
 * "Newton's method for large bound-constrained optimization problems",
 * SIAM J. Optim. 9(4), 1999; MINPACK-2 dtron).
 *
 * The reference (ExaAdmm.jl v0.6.1) calls this algorithm through the
 * third-party package ExaTron.jl (Project.toml:10,22, compat "3", no pinned
 * patch version, sources NOT present under /root/reference).  Call sites:
 *   src/models/acopf/acopf_auglag_linelimit_kernel_cpu.jl:104-106,116
 *   src/models/acopf/acopf_tron_linelimit_kernel.jl:103,114-116,122
 * This file restates the published algorithm for small dense problems
 * (n <= TRON_NMAX); the reverse-communication driver mirrors
 * acopf_tron_linelimit_kernel.jl:44-144.
 *
 * PARITY STATUS: primitive-level parity with ExaTron itself is UNPINNED (no
 * ExaTron sources or unit vectors available); end-to-end parity is pinned by
 * the reference's own known answers (test/algorithms/acopf_update_cpu.jl:28-172),
 * see tests/.  Each primitive is checked against an independent computation of
 * what the published algorithm defines (numpy/scipy factorisations and Krylov
 * solves, the Cauchy / projected-search inequalities, enumerated box-QP
 * minimisers, scipy.optimize on the bounded Rosenbrock function) in
 * tests/test_oracle_tron.py.
 */
#ifndef EXAADMM_ORACLE_TRON_DENSE_H
#define EXAADMM_ORACLE_TRON_DENSE_H

#define TRON_NMAX 8

/* task codes, same numbering as acopf_tron_linelimit_kernel.jl */
enum { TRON_START = 0, TRON_F = 1, TRON_GH = 2, TRON_NEWX = 3, TRON_CONV = 4, TRON_WARN = 10 };

typedef struct {
    /* persistent state of the dtron state machine (isave/dsave of MINPACK-2) */
    int work;      /* 1 COMPUTE, 2 EVALUATE, 3 NEWX */
    int iter;
    int iterscg;
    double fc, alphac, prered;
    double xc[TRON_NMAX];
    double s[TRON_NMAX];
} tron_state;

typedef double (*tron_eval_f)(const double *x, void *ctx);
/* fills g[n] and the full symmetric column-major A[n*n] */
typedef void (*tron_eval_gh)(const double *x, double *g, double *A, void *ctx);

typedef struct {
    int max_feval;   /* 500 */
    int max_minor;   /* 200 */
    double gtol;     /* 1e-6 */
    double frtol;    /* 1e-12 */
    double fatol;    /* 0 */
    double fmin;     /* -1e32 */
    double cgtol;    /* 0.1 */
} tron_options;

void tron_default_options(tron_options *o);

/* One trust-region Newton solve from x (feasible) ; returns status (0 ok, 1 = max_minor hit).
 * minor_iter/nfev are outputs. */
int tron_solve(int n, double *x, const double *xl, const double *xu,
               tron_eval_f ef, tron_eval_gh egh, void *ctx,
               const tron_options *opt, int *minor_iter, int *nfev);

/* exposed for unit tests */
double tron_dnrm2(int n, const double *x);
double tron_dgpnorm(int n, const double *x, const double *xl, const double *xu, const double *g);
void tron_dbreakpt(int n, const double *x, const double *xl, const double *xu, const double *w,
                   int *nbrpt, double *brptmin, double *brptmax);
void tron_dgpstep(int n, const double *x, const double *xl, const double *xu, double alpha,
                  const double *w, double *s);
double tron_dtrqsol(int n, const double *x, const double *p, double delta);
int tron_dicfs(int n, double alpha, const double *A, double *L);
void tron_dcauchy(int n, const double *x, const double *xl, const double *xu, const double *A,
                  const double *g, double delta, double *alpha, double *s);
int tron_dtrpcg(int n, const double *A, const double *g, double delta, const double *L,
                double tol, double stol, int itermax, double *w, int *iters);
void tron_dprsrch(int n, double *x, const double *xl, const double *xu, const double *A,
                  const double *g, double *w);
int tron_dspcg(int n, double *x, const double *xl, const double *xu, const double *A,
               const double *g, double delta, double rtol, double *s, int itermax, int *iters);
double tron_dtron(int n, double *x, const double *xl, const double *xu, double f, const double *g,
                  const double *A, double frtol, double fatol, double fmin, double cgtol,
                  int itermax, double delta, int *task, tron_state *st);

#endif
