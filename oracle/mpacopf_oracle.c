/*
 * ORACLE -- test infrastructure only (see mpacopf_oracle.h header).
 * CPU restatement of the reference's multi-period ACOPF (ModelMpacopf) ADMM steps.
 */
#define _POSIX_C_SOURCE 199309L
#include "mpacopf_oracle.h"
#include "tron_dense.h"
#include <math.h>
#include <string.h>
#include <time.h>

#define NVAR(g) (2 * (g)->ngen + 8 * (g)->nline)

/* mod.nvar (mpacopf_model.jl:99-104) */
int oracle_mp_nvar(const oracle_mp *m)
{
    const oracle_grid *g = &m->grids[0];
    return (m->T == 1) ? NVAR(g) : NVAR(g) + g->ngen;
}

/* init_solution! (mpacopf_init_solution_cpu.jl:1-20) */
void oracle_mp_init_solution(oracle_mp *m, double rho_pq, double rho_va)
{
    const int G = m->grids[0].ngen;
    for (int i = 0; i < m->T; i++) {
        oracle_init_solution(&m->grids[i], &m->states[i], rho_pq, rho_va);
        oracle_ramp *r = &m->ramps[i];
        double *vecs[] = {r->u_curr, r->l_curr, r->rd, r->rp, r->z_curr, r->z_prev, r->lz, r->Ax_plus_By, r->s_curr};
        for (unsigned k = 0; k < sizeof(vecs) / sizeof(vecs[0]); k++) memset(vecs[k], 0, sizeof(double) * G);
        for (int g = 0; g < G; g++) r->rho[g] = rho_pq;
        if (i > 0) {
            for (int g = 0; g < G; g++) {
                r->u_curr[g] = m->states[i - 1].v_curr[2 * g];
                r->s_curr[g] = m->states[i].u_curr[2 * g] - r->u_curr[g];
            }
        }
    }
}

/* admm_inner_prestep (mpacopf_admm_prepoststep_cpu.jl:24-38) */
void oracle_mp_inner_prestep(oracle_mp *m)
{
    const int G = m->grids[0].ngen;
    for (int i = 0; i < m->T; i++) oracle_inner_prestep(&m->grids[i], &m->states[i]);
    for (int i = 1; i < m->T; i++) memcpy(m->ramps[i].z_prev, m->ramps[i].z_curr, sizeof(double) * G);
}

/* ---------------------------------------------------------------------------
 * generator ramp subproblem callbacks (mpacopf_eval_generator_kernel_cpu.jl:1-101)
 * param rows (0-based): 0 lambda_p, 1 lambda_phat, 2 rho_p, 3 rho_phat, 4 pbar_t - z_p,
 * 5 pbar_{t-1} - z_phat, 6 multiplier mu, 7 penalty xi
 * ------------------------------------------------------------------------- */
double oracle_genramp_eval_f(const double *x, const double *param, double scale, double c2, double c1,
                             double c0, double baseMVA)
{
    double f = 0.0;
    f += c2 * ((x[0] * baseMVA) * (x[0] * baseMVA)) + c1 * (x[0] * baseMVA) + c0;
    f += param[0] * (x[0] - param[4]) + (0.5 * param[2]) * ((x[0] - param[4]) * (x[0] - param[4]));
    f += param[1] * (x[1] - param[5]) + (0.5 * param[3]) * ((x[1] - param[5]) * (x[1] - param[5]));
    f += param[6] * (x[0] - x[1] - x[2]) + (0.5 * param[7]) * ((x[0] - x[1] - x[2]) * (x[0] - x[1] - x[2]));
    return f * scale;
}

void oracle_genramp_eval_gh(const double *x, const double *param, double scale, double c2, double c1,
                            double c0, double baseMVA, double *g, double *A)
{
    (void)c0;
    const double M2 = baseMVA * baseMVA;
    g[0] = 2 * c2 * M2 * x[0] + c1 * baseMVA;
    g[0] += param[0] + param[2] * (x[0] - param[4]);
    g[0] += param[6] + param[7] * (x[0] - x[1] - x[2]);
    g[0] *= scale;
    g[1] = param[1] + param[3] * (x[1] - param[5]);
    g[1] += -(param[6] + param[7] * (x[0] - x[1] - x[2]));
    g[1] *= scale;
    g[2] = -(param[6] + param[7] * (x[0] - x[1] - x[2]));
    g[2] *= scale;

    /* symmetric 3x3, column-major (mpacopf_eval_generator_kernel_cpu.jl:52-101) */
    A[0] = scale * (2 * c2 * M2 + param[2] + param[7]);
    A[1] = A[3] = scale * (-param[7]);
    A[2] = A[6] = scale * (-param[7]);
    A[4] = scale * (param[3] + param[7]);
    A[5] = A[7] = scale * (param[7]);
    A[8] = scale * (param[7]);
}

typedef struct {
    const double *param;
    double scale, c2, c1, c0, baseMVA;
} gen_ctx;

static double gcb_f(const double *x, void *ctx)
{
    gen_ctx *c = (gen_ctx *)ctx;
    return oracle_genramp_eval_f(x, c->param, c->scale, c->c2, c->c1, c->c0, c->baseMVA);
}

static void gcb_gh(const double *x, double *g, double *A, void *ctx)
{
    gen_ctx *c = (gen_ctx *)ctx;
    oracle_genramp_eval_gh(x, c->param, c->scale, c->c2, c->c1, c->c0, c->baseMVA, g, A);
}

/* augmented-Lagrangian loop of one generator (mpacopf_auglag_generator_kernel_cpu.jl:58-122);
 * param[7] (xi) must be set on entry; param[6], param[7] are updated in place */
int oracle_genramp_solve(double *x, const double *xl, const double *xu, double *param, double scale,
                         double c2, double c1, double c0, double baseMVA, int max_auglag, double xi_max,
                         int *auglag_it, int *minor_it)
{
    tron_options opt;
    tron_default_options(&opt);
    gen_ctx ctx = {param, scale, c2, c1, c0, baseMVA};
    double xi = param[7];
    double eta = 1 / pow(xi, 0.1);
    int it = 0, minor_total = 0, terminate = 0, hit_limit = 0;
    while (!terminate) {
        it++;
        int minor = 0;
        tron_solve(3, x, xl, xu, gcb_f, gcb_gh, &ctx, &opt, &minor, 0);
        minor_total += minor;
        const double cviol = x[0] - x[1] - x[2];
        const double cnorm = fabs(cviol);
        if (cnorm <= eta) {
            if (cnorm <= 1e-6) {
                terminate = 1;
            } else {
                param[6] += xi * cviol;
                eta = eta / pow(xi, 0.9);
            }
        } else {
            xi = fmin(xi_max, xi * 10);
            eta = 1 / pow(xi, 0.1);
            param[7] = xi;
        }
        if (it >= max_auglag) {
            hit_limit = !terminate;
            terminate = 1;
        }
    }
    if (auglag_it) *auglag_it = it;
    if (minor_it) *minor_it = minor_total;
    return hit_limit;
}

/* auglag_generator_kernel (mpacopf_auglag_generator_kernel_cpu.jl:1-127), period i >= 1 (0-based),
 * called with scale = 1.0 and xi_max = par.mu_max (mpacopf_admm_update_x_cpu.jl:15-25) */
static void generator_ramp_update(const oracle_grid *g, oracle_state *s, oracle_ramp *r, const double *v_prev,
                                  const double *ramp_rate, int major_iter, int max_auglag, double xi_max)
{
    for (int I = 0; I < g->ngen; I++) {
        const int pg = 2 * I, qg = 2 * I + 1;
        double *param = r->gen_membuf + ORACLE_GEN_MEMBUF_ROWS * I;
        s->u_curr[qg] = fmax(g->qgmin[I], fmin(g->qgmax[I],
                             (-(s->l_curr[qg] + s->rho[qg] * (-s->v_curr[qg] + s->z_curr[qg]))) / s->rho[qg]));
        double x[3], xl[3], xu[3];
        xl[0] = xl[1] = g->pgmin[I];
        xu[0] = xu[1] = g->pgmax[I];
        xl[2] = -ramp_rate[I];
        xu[2] = ramp_rate[I];
        x[0] = fmin(xu[0], fmax(xl[0], s->u_curr[pg]));
        x[1] = fmin(xu[1], fmax(xl[1], r->u_curr[I]));
        x[2] = fmin(xu[2], fmax(xl[2], r->s_curr[I]));
        param[0] = s->l_curr[pg];
        param[1] = r->l_curr[I];
        param[2] = s->rho[pg];
        param[3] = r->rho[I];
        param[4] = s->v_curr[pg] - s->z_curr[pg];
        param[5] = v_prev[pg] - r->z_curr[I];
        if (major_iter <= 1) param[7] = 10.0;
        oracle_genramp_solve(x, xl, xu, param, 1.0, g->c2[I], g->c1[I], g->c0[I], g->baseMVA, max_auglag, xi_max, 0, 0);
        s->u_curr[pg] = x[0];
        r->u_curr[I] = x[1];
        r->s_curr[I] = x[2];
    }
}

/* admm_update_x (mpacopf_admm_update_x_cpu.jl:5-45) */
void oracle_mp_update_x(oracle_mp *m, const oracle_params *p, int major_iter)
{
    oracle_generator_update(&m->grids[0], &m->states[0]);
    for (int i = 1; i < m->T; i++)
        generator_ramp_update(&m->grids[i], &m->states[i], &m->ramps[i], m->states[i - 1].v_curr, m->ramp_rate,
                              major_iter, p->max_auglag, p->mu_max);
    for (int i = 0; i < m->T; i++) oracle_branch_update(&m->grids[i], p, &m->states[i], major_iter, &m->sub_info[i]);
}

/* bus_kernel_ramp (mpacopf_bus_kernel_cpu.jl:1-121): bus update of period i with the ramp copy
 * phat of period i+1 attached to the generators' real power */
static void bus_update_ramp(const oracle_grid *g, oracle_state *s, const oracle_ramp *r)
{
    const double *u = s->u_curr, *z = s->z_curr, *l = s->l_curr, *rho = s->rho;
    const double *r_u = r->u_curr, *r_z = r->z_curr, *r_l = r->l_curr, *r_rho = r->rho;
    double *v = s->v_curr;
    const int ls = 2 * g->ngen;
    for (int I = 0; I < g->nbus; I++) {
        double common_wi = 0.0, common_ti = 0.0;
        double inv_rhosum_pij_ji = 0.0, inv_rhosum_qij_ji = 0.0;
        double rhosum_wi = 0.0, rhosum_ti = 0.0;
        for (int k = g->FrStart[I]; k < g->FrStart[I + 1]; k++) {
            int p = ls + 8 * g->FrIdx[k];
            common_wi += l[p + 4] + rho[p + 4] * (u[p + 4] + z[p + 4]);
            common_ti += l[p + 6] + rho[p + 6] * (u[p + 6] + z[p + 6]);
            inv_rhosum_pij_ji += 1.0 / rho[p];
            inv_rhosum_qij_ji += 1.0 / rho[p + 1];
            rhosum_wi += rho[p + 4];
            rhosum_ti += rho[p + 6];
        }
        for (int k = g->ToStart[I]; k < g->ToStart[I + 1]; k++) {
            int p = ls + 8 * g->ToIdx[k];
            common_wi += l[p + 5] + rho[p + 5] * (u[p + 5] + z[p + 5]);
            common_ti += l[p + 7] + rho[p + 7] * (u[p + 7] + z[p + 7]);
            inv_rhosum_pij_ji += 1.0 / rho[p + 2];
            inv_rhosum_qij_ji += 1.0 / rho[p + 3];
            rhosum_wi += rho[p + 5];
            rhosum_ti += rho[p + 7];
        }
        common_wi /= rhosum_wi;

        double rhs1 = 0.0, rhs2 = 0.0, inv_rhosum_pg = 0.0, inv_rhosum_qg = 0.0;
        for (int k = g->GenStart[I]; k < g->GenStart[I + 1]; k++) {
            int gid = g->GenIdx[k], pg = 2 * gid;
            rhs1 += ((l[pg] + rho[pg] * (u[pg] + z[pg])) + (r_l[gid] + r_rho[gid] * (r_u[gid] + r_z[gid]))) /
                    (rho[pg] + r_rho[gid]);
            rhs2 += (u[pg + 1] + z[pg + 1]) + (l[pg + 1] / rho[pg + 1]);
            inv_rhosum_pg += 1.0 / (rho[pg] + r_rho[gid]);
            inv_rhosum_qg += 1.0 / rho[pg + 1];
        }
        rhs1 -= (g->Pd[I] / g->baseMVA);
        rhs2 -= (g->Qd[I] / g->baseMVA);
        for (int k = g->FrStart[I]; k < g->FrStart[I + 1]; k++) {
            int p = ls + 8 * g->FrIdx[k];
            rhs1 -= (u[p] + z[p]) + (l[p] / rho[p]);
            rhs2 -= (u[p + 1] + z[p + 1]) + (l[p + 1] / rho[p + 1]);
        }
        for (int k = g->ToStart[I]; k < g->ToStart[I + 1]; k++) {
            int p = ls + 8 * g->ToIdx[k];
            rhs1 -= (u[p + 2] + z[p + 2]) + (l[p + 2] / rho[p + 2]);
            rhs2 -= (u[p + 3] + z[p + 3]) + (l[p + 3] / rho[p + 3]);
        }
        rhs1 -= g->YshR[I] * common_wi;
        rhs2 += g->YshI[I] * common_wi;

        double A11 = (inv_rhosum_pg + inv_rhosum_pij_ji) + (g->YshR[I] * g->YshR[I] / rhosum_wi);
        double A12 = -g->YshR[I] * (g->YshI[I] / rhosum_wi);
        double A21 = A12;
        double A22 = (inv_rhosum_qg + inv_rhosum_qij_ji) + (g->YshI[I] * g->YshI[I] / rhosum_wi);
        double mu2 = (rhs2 - (A21 / A11) * rhs1) / (A22 - (A21 / A11) * A12);
        double mu1 = (rhs1 - A12 * mu2) / A11;
        double wi = common_wi + ((g->YshR[I] * mu1 - g->YshI[I] * mu2) / rhosum_wi);
        double ti = common_ti / rhosum_ti;

        for (int k = g->GenStart[I]; k < g->GenStart[I + 1]; k++) {
            int gid = g->GenIdx[k], pg = 2 * gid;
            v[pg] = ((l[pg] + rho[pg] * (u[pg] + z[pg])) + (r_l[gid] + r_rho[gid] * (r_u[gid] + r_z[gid])) - mu1) /
                    (rho[pg] + r_rho[gid]);
            v[pg + 1] = (u[pg + 1] + z[pg + 1]) + (l[pg + 1] - mu2) / rho[pg + 1];
        }
        for (int k = g->FrStart[I]; k < g->FrStart[I + 1]; k++) {
            int p = ls + 8 * g->FrIdx[k];
            v[p] = (u[p] + z[p]) + (l[p] + mu1) / rho[p];
            v[p + 1] = (u[p + 1] + z[p + 1]) + (l[p + 1] + mu2) / rho[p + 1];
            v[p + 4] = wi;
            v[p + 6] = ti;
        }
        for (int k = g->ToStart[I]; k < g->ToStart[I + 1]; k++) {
            int p = ls + 8 * g->ToIdx[k];
            v[p + 2] = (u[p + 2] + z[p + 2]) + (l[p + 2] + mu1) / rho[p + 2];
            v[p + 3] = (u[p + 3] + z[p + 3]) + (l[p + 3] + mu2) / rho[p + 3];
            v[p + 5] = wi;
            v[p + 7] = ti;
        }
    }
}

/* admm_update_xbar (mpacopf_admm_update_xbar_cpu.jl:1-20) */
void oracle_mp_update_xbar(oracle_mp *m)
{
    for (int i = 0; i < m->T - 1; i++) bus_update_ramp(&m->grids[i], &m->states[i], &m->ramps[i + 1]);
    oracle_bus_update(&m->grids[m->T - 1], &m->states[m->T - 1]);
}

/* admm_update_z (mpacopf_admm_update_z_cpu.jl:1-17) */
void oracle_mp_update_z(oracle_mp *m, double beta)
{
    const int G = m->grids[0].ngen;
    for (int i = 0; i < m->T; i++) oracle_update_z(&m->grids[i], &m->states[i], beta);
    for (int i = 1; i < m->T; i++) {
        oracle_ramp *r = &m->ramps[i];
        const double *v = m->states[i - 1].v_curr;
        for (int g = 0; g < G; g++)
            r->z_curr[g] = (-(r->lz[g] + r->l_curr[g] + r->rho[g] * (r->u_curr[g] - v[2 * g]))) / (beta + r->rho[g]);
    }
}

/* admm_update_l (mpacopf_admm_update_l_cpu.jl:1-16) */
void oracle_mp_update_l(oracle_mp *m, double beta)
{
    const int G = m->grids[0].ngen;
    for (int i = 0; i < m->T; i++) oracle_update_l(&m->grids[i], &m->states[i], beta);
    for (int i = 1; i < m->T; i++) {
        oracle_ramp *r = &m->ramps[i];
        for (int g = 0; g < G; g++) r->l_curr[g] = -(r->lz[g] + beta * r->z_curr[g]);
    }
}

/* admm_update_lz (mpacopf_admm_update_lz_cpu.jl:1-16) */
void oracle_mp_update_lz(oracle_mp *m, double beta, double max_mult)
{
    const int G = m->grids[0].ngen;
    for (int i = 0; i < m->T; i++) oracle_update_lz(&m->grids[i], &m->states[i], beta, max_mult);
    for (int i = 1; i < m->T; i++) {
        oracle_ramp *r = &m->ramps[i];
        for (int g = 0; g < G; g++) r->lz[g] = fmax(-max_mult, fmin(max_mult, r->lz[g] + (beta * r->z_curr[g])));
    }
}

static double norm2(const double *x, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s);
}

/* admm_update_residual (mpacopf_admm_update_residual_cpu.jl:1-55): per-period norms (period i >= 2
 * includes its ramp block), model norm = MAX over periods */
void oracle_mp_update_residual(oracle_mp *m, oracle_info *info)
{
    const int G = m->grids[0].ngen;
    info->primres = info->dualres = info->norm_z_curr = info->mismatch = 0.0;
    for (int i = 0; i < m->T; i++) oracle_update_residual(&m->grids[i], &m->states[i], &m->sub_info[i]);
    for (int i = 1; i < m->T; i++) {
        oracle_ramp *r = &m->ramps[i];
        const double *v = m->states[i - 1].v_curr;
        for (int g = 0; g < G; g++) {
            r->rp[g] = r->u_curr[g] - v[2 * g] + r->z_curr[g];
            r->rd[g] = r->z_curr[g] - r->z_prev[g];
            r->Ax_plus_By[g] = r->rp[g] - r->z_curr[g];
        }
        oracle_info *si = &m->sub_info[i];
        double t;
        t = norm2(r->rp, G);         si->primres = sqrt(si->primres * si->primres + t * t);
        t = norm2(r->rd, G);         si->dualres = sqrt(si->dualres * si->dualres + t * t);
        t = norm2(r->z_curr, G);     si->norm_z_curr = sqrt(si->norm_z_curr * si->norm_z_curr + t * t);
        t = norm2(r->Ax_plus_By, G); si->mismatch = sqrt(si->mismatch * si->mismatch + t * t);
    }
    for (int i = 0; i < m->T; i++) {
        info->primres = fmax(info->primres, m->sub_info[i].primres);
        info->dualres = fmax(info->dualres, m->sub_info[i].dualres);
        info->norm_z_curr = fmax(info->norm_z_curr, m->sub_info[i].norm_z_curr);
        info->mismatch = fmax(info->mismatch, m->sub_info[i].mismatch);
    }
}

/* admm_outer_prestep (mpacopf_admm_prepoststep_cpu.jl:1-22) */
double oracle_mp_norm_z_prev(oracle_mp *m)
{
    const int G = m->grids[0].ngen;
    double r = 0.0;
    for (int i = 0; i < m->T; i++) {
        double nz = oracle_norm_z(&m->grids[i], &m->states[i]);
        if (i > 0) {
            double t = norm2(m->ramps[i].z_curr, G);
            nz = sqrt(nz * nz + t * t);
        }
        m->sub_info[i].norm_z_prev = nz;
        r = fmax(r, nz);
    }
    return r;
}

/* admm_poststep (mpacopf_admm_prepoststep_cpu.jl:49-74): objective summed over periods */
double oracle_mp_objective(const oracle_mp *m)
{
    double obj = 0.0;
    for (int i = 0; i < m->T; i++) obj += oracle_objective(&m->grids[i], &m->states[i]);
    return obj;
}

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* admm_two_level (algorithms/admm_two_level.jl:1-88) on the multi-period model */
void oracle_mp_admm_two_level(oracle_mp *m, oracle_params *p, oracle_info *info)
{
    const double sqrt_d = sqrt((double)oracle_mp_nvar(m));
    const double OUTER_TOL = sqrt_d * p->outer_eps;
    memset(info, 0, sizeof(*info));
    for (int i = 0; i < m->T; i++) memset(&m->sub_info[i], 0, sizeof(oracle_info));
    info->mismatch = INFINITY;
    info->norm_z_prev = info->norm_z_curr = INFINITY;
    p->beta = p->initial_beta;
    info->status = 0;

    double t0 = now_sec();
    while (info->outer < p->outer_iterlim) {
        info->outer += 1;
        info->norm_z_prev = oracle_mp_norm_z_prev(m);
        info->inner = 0;
        while (info->inner < p->inner_iterlim) {
            info->inner += 1;
            info->cumul += 1;
            oracle_mp_inner_prestep(m);
            oracle_mp_update_x(m, p, info->inner);
            oracle_mp_update_xbar(m);
            oracle_mp_update_z(m, p->beta);
            oracle_mp_update_l(m, p->beta);
            oracle_mp_update_residual(m, info);
            info->eps_pri = sqrt_d / (2500.0 * info->outer);
            if (info->primres <= info->eps_pri) break;
        }
        if (info->mismatch <= OUTER_TOL) {
            info->status = 1;
            break;
        }
        oracle_mp_update_lz(m, p->beta, p->MAX_MULTIPLIER);
        if (info->norm_z_curr > p->theta * info->norm_z_prev) p->beta = fmin(p->inc_c * p->beta, 1e24);
    }
    info->time_overall = now_sec() - t0;
    info->objval = oracle_mp_objective(m);
}
