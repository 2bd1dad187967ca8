/*
 * ORACLE -- test infrastructure only (see acopf_oracle.h).  Nothing under oracle/ is linked
 * into, imported by, or shipped with the product library.
 *
 * Serial CPU restatement of the reference's multi-period ACOPF model (ModelMpacopf, SURVEY.md
 * §8 row f4): T single-period ACOPF models coupled through generator ramp constraints
 *   p_{t,g} - phat_{t-1,g} - s_{t,g} = 0,  |s| <= ramp_ratio * pgmax,   phat_{t-1,g} ~ pbar_{t-1,g}
 * Files followed (paths under /root/reference/src/models/mpacopf/):
 *   mpacopf_model.jl:1-112, mpacopf_init_solution_cpu.jl:1-20, mpacopf_admm_update_x_cpu.jl:5-45,
 *   mpacopf_auglag_generator_kernel_cpu.jl:1-127, mpacopf_eval_generator_kernel_cpu.jl:1-101,
 *   mpacopf_bus_kernel_cpu.jl:1-121, mpacopf_admm_update_{xbar,z,l,lz,residual}_cpu.jl,
 *   mpacopf_admm_prepoststep_cpu.jl:1-74, mpacopf_admm_increment.jl, interface/solve_mpacopf.jl.
 *
 * PARITY STATUS: pinned by the reference's case9 / 3-period known answers
 * (test/algorithms/mpacopf_update_cpu.jl: one-iteration u/v/z/l vectors per period, and
 * Solved / outer 20 / cumul 729 / objval 15901.48).  The reference's load profile files
 * (case9_onehour_60.Pd/.Qd, an Artifact) are NOT available offline: the profile is inferred
 * from the golden v vectors (a uniform scaling of the case9 loads per period, 2 fitted scalars);
 * see tests/golden/gen_mpacopf_case9.py.
 */
#ifndef EXAADMM_ORACLE_MPACOPF_H
#define EXAADMM_ORACLE_MPACOPF_H

#include "acopf_oracle.h"

#define ORACLE_GEN_MEMBUF_ROWS 8

/* mirrors SolutionRamping (mpacopf_model.jl:1-38); all vectors have ngen entries; v_curr of the
 * struct is never read on this path (the consensus partner is v_curr[pg] of the previous period) */
typedef struct {
    double *u_curr, *l_curr, *rho, *rd, *rp, *z_curr, *z_prev, *lz, *Ax_plus_By, *s_curr;
    double *gen_membuf; /* 8 x ngen column-major; rows 6,7 (0-based) = multiplier, penalty xi: persistent */
} oracle_ramp;

/* T periods: grids[t] differ in Pd/Qd only; ramps[0] is allocated but unused (as in the reference) */
typedef struct {
    int32_t T;
    const oracle_grid *grids;   /* T */
    oracle_state *states;       /* T */
    oracle_ramp *ramps;         /* T */
    const double *ramp_rate;    /* ngen: ramp_ratio * pgmax (acopf_model.jl:66-67) */
    oracle_info *sub_info;      /* T: per-period residual norms */
} oracle_mp;

int  oracle_mp_nvar(const oracle_mp *m);
void oracle_mp_init_solution(oracle_mp *m, double rho_pq, double rho_va);
void oracle_mp_inner_prestep(oracle_mp *m);
void oracle_mp_update_x(oracle_mp *m, const oracle_params *p, int major_iter);
void oracle_mp_update_xbar(oracle_mp *m);
void oracle_mp_update_z(oracle_mp *m, double beta);
void oracle_mp_update_l(oracle_mp *m, double beta);
void oracle_mp_update_lz(oracle_mp *m, double beta, double max_mult);
void oracle_mp_update_residual(oracle_mp *m, oracle_info *info);
double oracle_mp_norm_z_prev(oracle_mp *m);
double oracle_mp_objective(const oracle_mp *m);
void oracle_mp_admm_two_level(oracle_mp *m, oracle_params *p, oracle_info *info);

/* generator ramp subproblem, exposed for unit tests: x = (p_t, phat_{t-1}, s_t), param = 8 rows */
double oracle_genramp_eval_f(const double *x, const double *param, double scale, double c2, double c1,
                             double c0, double baseMVA);
void oracle_genramp_eval_gh(const double *x, const double *param, double scale, double c2, double c1,
                            double c0, double baseMVA, double *g, double *A);
int oracle_genramp_solve(double *x, const double *xl, const double *xu, double *param, double scale,
                         double c2, double c1, double c0, double baseMVA, int max_auglag, double xi_max,
                         int *auglag_it, int *minor_it);

#endif
