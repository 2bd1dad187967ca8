/*
 * ORACLE -- test infrastructure only (see acopf_oracle.h header).
 * CPU restatement of the reference's two-level ADMM ACOPF inner loop.
 */
#include "acopf_oracle.h"
#include "tron_dense.h"
#include <math.h>
#include <string.h>
#include <time.h>

#define NVAR(g) (2 * (g)->ngen + 8 * (g)->nline)
#define LINE_START(g) (2 * (g)->ngen)

void oracle_default_params(oracle_params *p)
{
    /* utils/environment.jl:43-75 */
    p->mu_max = 1e8;
    p->max_auglag = 50;
    p->initial_beta = 1e3;
    p->beta = 1e3;
    p->inc_c = 6.0;
    p->theta = 0.8;
    p->outer_eps = 2e-4;
    p->MAX_MULTIPLIER = 1e12;
    p->outer_iterlim = 20;
    p->inner_iterlim = 1000;
    p->scale = 1e-4;
    p->nthreads = 1;
}

/* ---------------------------------------------------------------------------
 * init_solution!  (models/acopf/acopf_init_solution_cpu.jl:1-45)
 * membuf fill     (models/acopf/acopf_model.jl:87-89)
 * pg bounds copy  (models/acopf/acopf_model.jl:61-64)
 * ------------------------------------------------------------------------- */
void oracle_init_solution(const oracle_grid *g, oracle_state *s, double rho_pq, double rho_va)
{
    const int nvar = NVAR(g);
    double *vecs[] = {s->u_curr, s->v_curr, s->l_curr, s->rd, s->rp,
                      s->z_curr, s->z_prev, s->lz, s->Ax_plus_By};
    for (unsigned k = 0; k < sizeof(vecs) / sizeof(vecs[0]); k++)
        memset(vecs[k], 0, sizeof(double) * nvar);
    for (int i = 0; i < nvar; i++) s->rho[i] = rho_pq;

    for (int i = 0; i < g->ngen; i++) {
        s->v_curr[2 * i] = 0.5 * (g->pgmin[i] + g->pgmax[i]);
        s->v_curr[2 * i + 1] = 0.5 * (g->qgmin[i] + g->qgmax[i]);
        s->pgmin_curr[i] = g->pgmin[i];
        s->pgmax_curr[i] = g->pgmax[i];
    }
    for (int l = 0; l < g->nline; l++) {
        int fr = g->brBusIdx[2 * l], to = g->brBusIdx[2 * l + 1];
        double wij0 = (g->Vmax[fr] * g->Vmax[fr] + g->Vmin[fr] * g->Vmin[fr]) / 2;
        double wji0 = (g->Vmax[to] * g->Vmax[to] + g->Vmin[to] * g->Vmin[to]) / 2;
        double wR0 = sqrt(wij0 * wji0);
        int p = LINE_START(g) + 8 * l;
        s->v_curr[p] = g->YffR[l] * wij0 + g->YftR[l] * wR0;
        s->v_curr[p + 1] = -g->YffI[l] * wij0 - g->YftI[l] * wR0;
        s->v_curr[p + 2] = g->YttR[l] * wji0 + g->YtfR[l] * wR0;
        s->v_curr[p + 3] = -g->YttI[l] * wji0 - g->YtfI[l] * wR0;
        s->v_curr[p + 4] = wij0;
        s->v_curr[p + 5] = wji0;
        s->v_curr[p + 6] = 0.0;
        s->v_curr[p + 7] = 0.0;
        for (int k = 4; k < 8; k++) s->rho[p + k] = rho_va;
    }
    memset(s->membuf, 0, sizeof(double) * ORACLE_MEMBUF_ROWS * g->nline);
    for (int l = 0; l < g->nline; l++) s->membuf[28 + ORACLE_MEMBUF_ROWS * l] = g->rateA[l];
}

/* ---------------------------------------------------------------------------
 * generator_kernel_two_level (models/acopf/acopf_generator_kernel_cpu.jl:1-20),
 * called with pgmin_curr/pgmax_curr (:28-30)
 * ------------------------------------------------------------------------- */
void oracle_generator_update(const oracle_grid *g, oracle_state *s)
{
    const double M = g->baseMVA;
    for (int i = 0; i < g->ngen; i++) {
        int pg = 2 * i, qg = 2 * i + 1;
        double t = (-(g->c1[i] * M + s->l_curr[pg] + s->rho[pg] * (-s->v_curr[pg] + s->z_curr[pg]))) /
                   (2 * g->c2[i] * (M * M) + s->rho[pg]);
        s->u_curr[pg] = fmax(s->pgmin_curr[i], fmin(s->pgmax_curr[i], t));
        t = (-(s->l_curr[qg] + s->rho[qg] * (-s->v_curr[qg] + s->z_curr[qg]))) / s->rho[qg];
        s->u_curr[qg] = fmax(g->qgmin[i], fmin(g->qgmax[i], t));
    }
}

/* ---------------------------------------------------------------------------
 * Branch augmented-Lagrangian callbacks.
 * Mathematics of models/acopf/acopf_eval_linelimit_kernel_cpu.jl:1-46 (f),
 * :48-172 (gradient), :174-605 (Hessian), restated through the chain rule on
 * the four flow functions instead of term-by-term expansion:
 *   phi_k(vi,vj,th) = a_k vi^2 + b_k vj^2 + vi vj (al_k cos th + be_k sin th),
 *   th = ti - tj, k = pij,qij,pji,qji.
 * param (0-based rows of membuf): 0-7 lambda, 8-15 rho, 16-23 (v - z),
 * 24,25 line-limit multipliers, 26 penalty mu.
 * Y = [YffR,YffI,YftR,YftI,YttR,YttI,YtfR,YtfI].
 * ------------------------------------------------------------------------- */
typedef struct {
    double a[4], b[4], al[4], be[4];
} flow_coef;

static void flow_coefficients(const double *Y, flow_coef *c)
{
    const double YffR = Y[0], YffI = Y[1], YftR = Y[2], YftI = Y[3];
    const double YttR = Y[4], YttI = Y[5], YtfR = Y[6], YtfI = Y[7];
    c->a[0] = YffR;  c->b[0] = 0.0;   c->al[0] = YftR;  c->be[0] = YftI;
    c->a[1] = -YffI; c->b[1] = 0.0;   c->al[1] = -YftI; c->be[1] = YftR;
    c->a[2] = 0.0;   c->b[2] = YttR;  c->al[2] = YtfR;  c->be[2] = -YtfI;
    c->a[3] = 0.0;   c->b[3] = -YttI; c->al[3] = -YtfI; c->be[3] = -YtfR;
}

static void branch_flows(const double *x, const double *Y, double *flow)
{
    /* acopf_eval_linelimit_kernel_cpu.jl:11-16 */
    double vi_vj_cos = x[0] * x[1] * cos(x[2] - x[3]);
    double vi_vj_sin = x[0] * x[1] * sin(x[2] - x[3]);
    flow[0] = Y[0] * (x[0] * x[0]) + Y[2] * vi_vj_cos + Y[3] * vi_vj_sin;
    flow[1] = -Y[1] * (x[0] * x[0]) - Y[3] * vi_vj_cos + Y[2] * vi_vj_sin;
    flow[2] = Y[4] * (x[1] * x[1]) + Y[6] * vi_vj_cos - Y[7] * vi_vj_sin;
    flow[3] = -Y[5] * (x[1] * x[1]) - Y[7] * vi_vj_cos - Y[6] * vi_vj_sin;
}

double oracle_branch_eval_f(const double *x, const double *param, double scale, const double *Y)
{
    /* acopf_eval_linelimit_kernel_cpu.jl:18-43, same accumulation order */
    double flow[4];
    branch_flows(x, Y, flow);
    double y[8] = {flow[0], flow[1], flow[2], flow[3], x[0] * x[0], x[1] * x[1], x[2], x[3]};
    double f = 0.0;
    for (int k = 0; k < 8; k++) f += param[k] * y[k];
    for (int k = 0; k < 8; k++) {
        double d = y[k] - param[16 + k];
        f += 0.5 * (param[8 + k] * (d * d));
    }
    double c1 = flow[0] * flow[0] + flow[1] * flow[1] + x[4];
    double c2 = flow[2] * flow[2] + flow[3] * flow[3] + x[5];
    f += param[24] * c1;
    f += param[25] * c2;
    f += 0.5 * (param[26] * (c1 * c1));
    f += 0.5 * (param[26] * (c2 * c2));
    return f * scale;
}

void oracle_branch_eval_gh(const double *x, const double *param, double scale, const double *Y,
                           double *g, double *A)
{
    const int n = 6;
    flow_coef c;
    flow_coefficients(Y, &c);
    const double vi = x[0], vj = x[1];
    const double cs = cos(x[2] - x[3]), sn = sin(x[2] - x[3]);
    const double mu = param[26];

    double phi[4], J[4][4], H[4][4][4];
    for (int k = 0; k < 4; k++) {
        double E = c.al[k] * cs + c.be[k] * sn;   /* al cos + be sin */
        double F = -c.al[k] * sn + c.be[k] * cs;  /* d/dth of E */
        phi[k] = c.a[k] * vi * vi + c.b[k] * vj * vj + vi * vj * E;
        J[k][0] = 2 * c.a[k] * vi + vj * E;
        J[k][1] = 2 * c.b[k] * vj + vi * E;
        J[k][2] = vi * vj * F;
        J[k][3] = -J[k][2];
        memset(H[k], 0, sizeof(H[k]));
        H[k][0][0] = 2 * c.a[k];
        H[k][1][1] = 2 * c.b[k];
        H[k][0][1] = H[k][1][0] = E;
        H[k][0][2] = H[k][2][0] = vj * F;
        H[k][0][3] = H[k][3][0] = -vj * F;
        H[k][1][2] = H[k][2][1] = vi * F;
        H[k][1][3] = H[k][3][1] = -vi * F;
        H[k][2][2] = -vi * vj * E;
        H[k][3][3] = -vi * vj * E;
        H[k][2][3] = H[k][3][2] = vi * vj * E;
    }
    const double c1 = (phi[0] * phi[0] + phi[1] * phi[1]) + x[4];
    const double c2 = (phi[2] * phi[2] + phi[3] * phi[3]) + x[5];
    const double m1 = param[24] + mu * c1; /* d/dc1 of the penalty terms */
    const double m2 = param[25] + mu * c2;

    /* w_k: total first-order weight on grad phi_k ; r_k: weight on the outer product */
    double w[4], r[4];
    for (int k = 0; k < 4; k++) {
        double mk = (k < 2) ? m1 : m2;
        w[k] = param[k] + param[8 + k] * (phi[k] - param[16 + k]) + 2 * phi[k] * mk;
        r[k] = param[8 + k] + 2 * mk;
    }
    /* gradients of c1, c2 w.r.t. (vi,vj,ti,tj) */
    double d1[4], d2[4];
    for (int i = 0; i < 4; i++) {
        d1[i] = 2 * (phi[0] * J[0][i] + phi[1] * J[1][i]);
        d2[i] = 2 * (phi[2] * J[2][i] + phi[3] * J[3][i]);
    }

    double gy[4];
    for (int i = 0; i < 4; i++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += w[k] * J[k][i];
        gy[i] = s;
    }
    const double wvi = param[4] + param[12] * (vi * vi - param[20]);
    const double wvj = param[5] + param[13] * (vj * vj - param[21]);
    gy[0] += wvi * (2 * vi);
    gy[1] += wvj * (2 * vj);
    gy[2] += param[6] + param[14] * (x[2] - param[22]);
    gy[3] += param[7] + param[15] * (x[3] - param[23]);

    g[0] = scale * gy[0];
    g[1] = scale * gy[1];
    g[2] = scale * gy[2];
    g[3] = scale * gy[3];
    g[4] = scale * m1;
    g[5] = scale * m2;

    double Hy[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j <= i; j++) { /* lower triangle, mirrored: exactly symmetric */
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += w[k] * H[k][i][j] + r[k] * J[k][i] * J[k][j];
            s += mu * (d1[i] * d1[j] + d2[i] * d2[j]);
            Hy[i][j] = s;
            Hy[j][i] = s;
        }
    Hy[0][0] += 2 * wvi + 4 * param[12] * vi * vi;
    Hy[1][1] += 2 * wvj + 4 * param[13] * vj * vj;
    Hy[2][2] += param[14];
    Hy[3][3] += param[15];

    for (int i = 0; i < n * n; i++) A[i] = 0.0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) A[i + n * j] = scale * Hy[i][j];
    for (int i = 0; i < 4; i++) {
        A[4 + n * i] = A[i + n * 4] = scale * (mu * d1[i]);
        A[5 + n * i] = A[i + n * 5] = scale * (mu * d2[i]);
    }
    A[4 + n * 4] = scale * mu;
    A[5 + n * 5] = scale * mu;
}

typedef struct {
    const double *param;
    const double *Y;
    double scale;
} branch_ctx;

static double cb_f(const double *x, void *ctx)
{
    branch_ctx *b = (branch_ctx *)ctx;
    return oracle_branch_eval_f(x, b->param, b->scale, b->Y);
}

static void cb_gh(const double *x, double *g, double *A, void *ctx)
{
    branch_ctx *b = (branch_ctx *)ctx;
    oracle_branch_eval_gh(x, b->param, b->scale, b->Y, g, A);
}

/* ---------------------------------------------------------------------------
 * Augmented-Lagrangian loop on one branch
 * (models/acopf/acopf_auglag_linelimit_kernel_cpu.jl:97-154).  param[26] (mu)
 * must be set on entry; param[24..26] are updated in place.
 * ------------------------------------------------------------------------- */
int oracle_branch_solve(double *x, const double *xl, const double *xu, double *param,
                        const double *Y, double scale, int max_auglag, double mu_max,
                        int *auglag_it, int *minor_it)
{
    tron_options opt;
    tron_default_options(&opt);
    branch_ctx ctx = {param, Y, scale};

    double mu = param[26];
    double eta = 1 / pow(mu, 0.1);
    int it = 0, minor_total = 0, terminate = 0, hit_limit = 0;

    while (!terminate) {
        it++;
        int minor = 0;
        tron_solve(6, x, xl, xu, cb_f, cb_gh, &ctx, &opt, &minor, 0);
        minor_total += minor;

        double flow[4];
        branch_flows(x, Y, flow);
        double cviol1 = flow[0] * flow[0] + flow[1] * flow[1] + x[4];
        double cviol2 = flow[2] * flow[2] + flow[3] * flow[3] + x[5];
        double cnorm = fmax(fabs(cviol1), fabs(cviol2));

        if (cnorm <= eta) {
            if (cnorm <= 1e-6) {
                terminate = 1;
            } else {
                param[24] += mu * cviol1;
                param[25] += mu * cviol2;
                eta = eta / pow(mu, 0.9);
            }
        } else {
            mu = fmin(mu_max, mu * 10);
            eta = 1 / pow(mu, 0.1);
            param[26] = mu;
        }
        if (it >= max_auglag) {
            hit_limit = !terminate;
            terminate = 1;
        }
    }
    param[26] = mu;
    *auglag_it = it;
    *minor_it = minor_total;
    return hit_limit;
}

/* ---------------------------------------------------------------------------
 * auglag_linelimit_two_level_alternative
 * (models/acopf/acopf_auglag_linelimit_kernel_cpu.jl:1-172)
 * ------------------------------------------------------------------------- */
void oracle_branch_update(const oracle_grid *g, const oracle_params *p, oracle_state *s,
                          int major_iter, oracle_info *info)
{
    double sum_auglag = 0.0, sum_minor = 0.0;
    long long tot_auglag = 0, tot_minor = 0;
    const int nline = g->nline;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(p->nthreads > 1 ? p->nthreads : 1) \
    reduction(+ : sum_auglag, sum_minor, tot_auglag, tot_minor) if (p->nthreads > 1)
#endif
    for (int I = 0; I < nline; I++) {
        double x[6], xl[6], xu[6];
        double *param = s->membuf + ORACLE_MEMBUF_ROWS * I;
        const double Y[8] = {g->YffR[I], g->YffI[I], g->YftR[I], g->YftI[I],
                             g->YttR[I], g->YttI[I], g->YtfR[I], g->YtfI[I]};
        const int pij = LINE_START(g) + 8 * I;
        double *u = s->u_curr;

        xl[0] = g->FrVmBound[2 * I]; xu[0] = g->FrVmBound[2 * I + 1];
        xl[1] = g->ToVmBound[2 * I]; xu[1] = g->ToVmBound[2 * I + 1];
        xl[2] = g->FrVaBound[2 * I]; xu[2] = g->FrVaBound[2 * I + 1];
        xl[3] = g->ToVaBound[2 * I]; xu[3] = g->ToVaBound[2 * I + 1];
        xl[4] = -param[28]; xu[4] = 0.0;
        xl[5] = -param[28]; xu[5] = 0.0;

        x[0] = fmin(xu[0], fmax(xl[0], sqrt(u[pij + 4])));
        x[1] = fmin(xu[1], fmax(xl[1], sqrt(u[pij + 5])));
        x[2] = fmin(xu[2], fmax(xl[2], u[pij + 6]));
        x[3] = fmin(xu[3], fmax(xl[3], u[pij + 7]));
        x[4] = fmin(xu[4], fmax(xl[4], -(u[pij] * u[pij] + u[pij + 1] * u[pij + 1])));
        x[5] = fmin(xu[5], fmax(xl[5], -(u[pij + 2] * u[pij + 2] + u[pij + 3] * u[pij + 3])));

        for (int k = 0; k < 8; k++) {
            param[k] = s->l_curr[pij + k];
            param[8 + k] = s->rho[pij + k];
            param[16 + k] = s->v_curr[pij + k] - s->z_curr[pij + k];
        }
        if (major_iter == 1) param[26] = 10.0;

        int it = 0, minor = 0;
        oracle_branch_solve(x, xl, xu, param, Y, p->scale, p->max_auglag, p->mu_max, &it, &minor);

        double flow[4];
        branch_flows(x, Y, flow);
        u[pij] = flow[0];
        u[pij + 1] = flow[1];
        u[pij + 2] = flow[2];
        u[pij + 3] = flow[3];
        u[pij + 4] = x[0] * x[0];
        u[pij + 5] = x[1] * x[1];
        u[pij + 6] = x[2];
        u[pij + 7] = x[3];

        sum_auglag += it;
        sum_minor += (double)minor / it;
        tot_auglag += it;
        tot_minor += minor;
    }
    if (info) {
        info->avg_auglag_it = nline ? sum_auglag / nline : 0.0;
        info->avg_minor_it = nline ? sum_minor / nline : 0.0;
        info->total_auglag_it += tot_auglag;
        info->total_minor_it += tot_minor;
    }
}

/* ---------------------------------------------------------------------------
 * bus_kernel_two_level_alternative (models/acopf/acopf_bus_kernel_cpu.jl:1-116)
 * ------------------------------------------------------------------------- */
void oracle_bus_update(const oracle_grid *g, oracle_state *s)
{
    const double *u = s->u_curr, *z = s->z_curr, *l = s->l_curr, *rho = s->rho;
    double *v = s->v_curr;
    const int ls = LINE_START(g);
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
            int pg = 2 * g->GenIdx[k];
            rhs1 += (u[pg] + z[pg]) + (l[pg] / rho[pg]);
            rhs2 += (u[pg + 1] + z[pg + 1]) + (l[pg + 1] / rho[pg + 1]);
            inv_rhosum_pg += 1.0 / rho[pg];
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
            int pg = 2 * g->GenIdx[k];
            v[pg] = (u[pg] + z[pg]) + (l[pg] - mu1) / rho[pg];
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

/* admm_inner_prestep (models/acopf/acopf_admm_prepoststep_cpu.jl:17-25) */
void oracle_inner_prestep(const oracle_grid *g, oracle_state *s)
{
    memcpy(s->z_prev, s->z_curr, sizeof(double) * NVAR(g));
}

/* admm_update_z (models/acopf/acopf_admm_update_z_cpu.jl:10) */
void oracle_update_z(const oracle_grid *g, oracle_state *s, double beta)
{
    const int n = NVAR(g);
    for (int i = 0; i < n; i++)
        s->z_curr[i] = (-(s->lz[i] + s->l_curr[i] + s->rho[i] * (s->u_curr[i] - s->v_curr[i]))) /
                       (beta + s->rho[i]);
}

/* admm_update_l (models/acopf/acopf_admm_update_l_cpu.jl:10) */
void oracle_update_l(const oracle_grid *g, oracle_state *s, double beta)
{
    const int n = NVAR(g);
    for (int i = 0; i < n; i++) s->l_curr[i] = -(s->lz[i] + beta * s->z_curr[i]);
}

/* admm_update_lz (models/acopf/acopf_admm_update_lz_cpu.jl:10) */
void oracle_update_lz(const oracle_grid *g, oracle_state *s, double beta, double max_mult)
{
    const int n = NVAR(g);
    for (int i = 0; i < n; i++)
        s->lz[i] = fmax(-max_mult, fmin(max_mult, s->lz[i] + (beta * s->z_curr[i])));
}

static double norm2(const double *x, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s);
}

double oracle_norm_z(const oracle_grid *g, const oracle_state *s)
{
    return norm2(s->z_curr, NVAR(g));
}

/* objective as in admm_poststep (models/acopf/acopf_admm_prepoststep_cpu.jl:42-45) */
double oracle_objective(const oracle_grid *g, const oracle_state *s)
{
    double obj = 0.0;
    for (int i = 0; i < g->ngen; i++) {
        double pg = g->baseMVA * s->u_curr[2 * i];
        obj += g->c2[i] * (pg * pg) + g->c1[i] * pg + g->c0[i];
    }
    return obj;
}

/* admm_update_residual (models/acopf/acopf_admm_update_residual_cpu.jl:13-38);
 * info.auglag (quirk: sums over 1:1 only, printed only) is not reproduced. */
void oracle_update_residual(const oracle_grid *g, oracle_state *s, oracle_info *info)
{
    const int n = NVAR(g);
    for (int i = 0; i < n; i++) {
        s->rp[i] = s->u_curr[i] - s->v_curr[i] + s->z_curr[i];
        s->rd[i] = s->z_curr[i] - s->z_prev[i];
        s->Ax_plus_By[i] = s->rp[i] - s->z_curr[i];
    }
    info->primres = norm2(s->rp, n);
    info->dualres = norm2(s->rd, n);
    info->norm_z_curr = norm2(s->z_curr, n);
    info->mismatch = norm2(s->Ax_plus_By, n);
    info->objval = oracle_objective(g, s);
}

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------------------------------------------------------------------
 * admm_two_level (algorithms/admm_two_level.jl:1-88)
 * ------------------------------------------------------------------------- */
void oracle_admm_two_level(const oracle_grid *g, oracle_params *p, oracle_state *s, oracle_info *info)
{
    const double sqrt_d = sqrt((double)NVAR(g));
    const double OUTER_TOL = sqrt_d * p->outer_eps;

    memset(info, 0, sizeof(*info));
    info->mismatch = INFINITY;
    info->norm_z_prev = info->norm_z_curr = INFINITY;
    p->beta = p->initial_beta;
    info->status = 0;

    double t0 = now_sec();
    while (info->outer < p->outer_iterlim) {
        info->outer += 1;                          /* admm_increment_outer */
        info->norm_z_prev = oracle_norm_z(g, s);   /* admm_outer_prestep */
        info->inner = 0;                           /* admm_increment_reset_inner */
        while (info->inner < p->inner_iterlim) {
            info->inner += 1;                      /* admm_increment_inner */
            info->cumul += 1;
            oracle_inner_prestep(g, s);
            oracle_generator_update(g, s);         /* admm_update_x */
            oracle_branch_update(g, p, s, info->inner, info);
            oracle_bus_update(g, s);               /* admm_update_xbar */
            oracle_update_z(g, s, p->beta);
            oracle_update_l(g, s, p->beta);
            oracle_update_residual(g, s, info);
            info->eps_pri = sqrt_d / (2500.0 * info->outer);
            if (info->primres <= info->eps_pri) break;
        }
        if (info->mismatch <= OUTER_TOL) {
            info->status = 1;
            break;
        }
        oracle_update_lz(g, s, p->beta, p->MAX_MULTIPLIER);
        if (info->norm_z_curr > p->theta * info->norm_z_prev)
            p->beta = fmin(p->inc_c * p->beta, 1e24);
    }
    info->time_overall = now_sec() - t0;
    info->objval = oracle_objective(g, s);         /* admm_poststep */
}
