/*
 * ORACLE -- test infrastructure only.  Nothing under oracle/ is linked into,
 * imported by, or shipped with the product library (exaadmm.jl_amd/).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Serial CPU restatement of the ExaAdmm.jl (v0.6.1) two-level ADMM ACOPF
 * inner loop, CPU path.  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/src).  Layout is the reference's:
 * vectors of length nvar = 2*ngen + 8*nline, generators first [pg,qg], then
 * per branch [pij,qij,pji,qji,wi,wj,ti,tj] (models/acopf/acopf_model.jl:55-58);
 * indices are 0-based here (reference: 1-based).
 *
 * PARITY STATUS: pinned end-to-end by the reference's case9 known answers
 * (test/algorithms/acopf_update_cpu.jl:28-172: one-iteration u/v/z/l vectors,
 * Solved/outer 20/cumul 705/objval 5303.435).  The TRON primitives come from
 * ExaTron.jl, absent from /root/reference: primitive-level parity UNPINNED.
 */
#ifndef EXAADMM_ORACLE_ACOPF_H
#define EXAADMM_ORACLE_ACOPF_H

#include <stdint.h>

#define ORACLE_MEMBUF_ROWS 31

typedef struct {
    int32_t ngen, nline, nbus;
    double baseMVA;
    /* generators (utils/opfdata.jl:542-579) */
    const double *pgmin, *pgmax, *qgmin, *qgmax, *c2, *c1, *c0;
    /* branches (utils/opfdata.jl:417-483, 694-756) */
    const double *YffR, *YffI, *YftR, *YftI, *YttR, *YttI, *YtfR, *YtfI;
    const double *FrVmBound, *ToVmBound, *FrVaBound, *ToVaBound; /* 2*nline, (lo,hi) */
    const double *rateA;
    const int32_t *brBusIdx; /* 2*nline (from,to), 0-based */
    /* buses (utils/opfdata.jl:610-652) */
    const double *Pd, *Qd, *Vmin, *Vmax, *YshR, *YshI;
    const int32_t *FrStart, *FrIdx, *ToStart, *ToIdx, *GenStart, *GenIdx; /* CSR, 0-based */
} oracle_grid;

/* mirrors Parameters (utils/environment.jl:6-76); only fields the path reads */
typedef struct {
    double mu_max;          /* 1e8 */
    int32_t max_auglag;     /* 50 */
    double initial_beta;    /* 1e3 */
    double beta;            /* running */
    double inc_c;           /* 6 */
    double theta;           /* 0.8 */
    double outer_eps;       /* 2e-4 */
    double MAX_MULTIPLIER;  /* 1e12 */
    int32_t outer_iterlim;  /* 20 */
    int32_t inner_iterlim;  /* 1000 */
    double scale;           /* 1e-4 */
    int32_t nthreads;       /* 1 = serial like the reference; >1 = OpenMP over components */
} oracle_params;

/* mirrors Solution (utils/environment.jl:177-226); unused *_prev vectors omitted */
typedef struct {
    double *u_curr, *v_curr, *l_curr, *rho, *rd, *rp, *z_curr, *z_prev, *lz, *Ax_plus_By;
    double *membuf;     /* 31 x nline column-major (models/acopf/acopf_model.jl:87-89) */
    double *pgmin_curr, *pgmax_curr; /* models/acopf/acopf_model.jl:61-64 */
} oracle_state;

/* mirrors IterationInformation (utils/environment.jl:328-358) */
typedef struct {
    int32_t status; /* 0 IterationLimit, 1 Solved */
    int32_t inner, outer, cumul;
    double objval, primres, dualres, mismatch, norm_z_curr, norm_z_prev, eps_pri;
    double time_overall;
    double avg_auglag_it, avg_minor_it; /* last x-update */
    int64_t total_auglag_it, total_minor_it; /* summed over all branch solves */
} oracle_info;

void oracle_default_params(oracle_params *p);
void oracle_init_solution(const oracle_grid *g, oracle_state *s, double rho_pq, double rho_va);
void oracle_generator_update(const oracle_grid *g, oracle_state *s);
void oracle_branch_update(const oracle_grid *g, const oracle_params *p, oracle_state *s,
                          int major_iter, oracle_info *info);
void oracle_bus_update(const oracle_grid *g, oracle_state *s);
void oracle_inner_prestep(const oracle_grid *g, oracle_state *s);
void oracle_update_z(const oracle_grid *g, oracle_state *s, double beta);
void oracle_update_l(const oracle_grid *g, oracle_state *s, double beta);
void oracle_update_lz(const oracle_grid *g, oracle_state *s, double beta, double max_mult);
void oracle_update_residual(const oracle_grid *g, oracle_state *s, oracle_info *info);
double oracle_norm_z(const oracle_grid *g, const oracle_state *s);
double oracle_objective(const oracle_grid *g, const oracle_state *s);
void oracle_admm_two_level(const oracle_grid *g, oracle_params *p, oracle_state *s, oracle_info *info);

/* branch callbacks, exposed for unit tests (finite-difference checks) */
double oracle_branch_eval_f(const double *x, const double *param, double scale, const double *Y);
void oracle_branch_eval_gh(const double *x, const double *param, double scale, const double *Y,
                           double *g, double *A);
int oracle_branch_solve(double *x, const double *xl, const double *xu, double *param,
                        const double *Y, double scale, int max_auglag, double mu_max,
                        int *auglag_it, int *minor_it);

#endif
