// Generator ramp subproblem of the multi-period ACOPF model, one generator
// of one period t >= 2 per GPU lane:
//     min over x = (p_t, phat_{t-1}, s_t), box bounds,
//         cost(p_t) + consensus terms of p_t and phat_{t-1} + augmented Lagrangian of p_t - phat_{t-1} - s_t = 0
//
// Reference behaviour reproduced (paths under /root/reference/src/models/mpacopf/):
//   mpacopf_eval_generator_kernel_cpu.jl:1-101      objective, gradient, dense 3x3 Hessian
//   mpacopf_auglag_generator_kernel_cpu.jl:1-127    augmented-Lagrangian loop (called with scale = 1)
//   mpacopf_tron_generator_kernel.jl                TRON driver constants (same as the branch problem)
// The TRON code is the N-templated lane code of tron_device.h instantiated with N = 3.
#pragma once
#include "branch_solver.h"

namespace exa {

// param rows of the reference's gen_membuf: 1 lambda_p, 2 lambda_phat, 3 rho_p, 4 rho_phat,
// 5 pbar_t - z_p, 6 pbar_{t-1} - z_phat, 7 multiplier mu (persistent), 8 penalty xi (persistent)
struct GenRampData {
    double lam_p, lam_h, rho_p, rho_h, xt_p, xt_h, mu, xi;
    double c2, c1, c0, M;
};

EXA_HD double genramp_f(const GenRampData& d, const double (&x)[3])
{
    double f = 0.0;
    const double pm = x[0] * d.M;
    f += d.c2 * (pm * pm) + d.c1 * pm + d.c0;
    const double e1 = x[0] - d.xt_p, e2 = x[1] - d.xt_h, cv = x[0] - x[1] - x[2];
    f += d.lam_p * e1 + (0.5 * d.rho_p) * (e1 * e1);
    f += d.lam_h * e2 + (0.5 * d.rho_h) * (e2 * e2);
    f += d.mu * cv + (0.5 * d.xi) * (cv * cv);
    return f;  // scale = 1 (mpacopf_admm_update_x_cpu.jl:16)
}

EXA_HD void genramp_g(const GenRampData& d, const double (&x)[3], double (&g)[3])
{
    const double cv = x[0] - x[1] - x[2];
    double g0 = 2 * d.c2 * (d.M * d.M) * x[0] + d.c1 * d.M;
    g0 += d.lam_p + d.rho_p * (x[0] - d.xt_p);
    g0 += d.mu + d.xi * cv;
    double g1 = d.lam_h + d.rho_h * (x[1] - d.xt_h);
    g1 += -(d.mu + d.xi * cv);
    g[0] = g0;
    g[1] = g1;
    g[2] = -(d.mu + d.xi * cv);
}

template <class MA>
EXA_HD void genramp_h(const GenRampData& d, MA& A)
{
    A.lo(0, 0) = 2 * d.c2 * (d.M * d.M) + d.rho_p + d.xi;
    A.lo(1, 0) = -d.xi;
    A.lo(2, 0) = -d.xi;
    A.lo(1, 1) = d.rho_h + d.xi;
    A.lo(2, 1) = d.xi;
    A.lo(2, 2) = d.xi;
}

// Augmented-Lagrangian x TRON solve of one generator; d.mu, d.xi are updated.  Same flattened
// tick loop as branch_solve (one loop, every lane in the same phase).
EXA_HD void genramp_solve(GenRampData& d, const BranchOptions& o, const MuPowTable& pw, double (&x)[3],
                          double (&xl)[3], double (&xu)[3], int& auglag_it, int& minor_it)
{
    TronState<3> st;
    SymMat<3> A, L;
    double g[3];
    double xi = d.xi;
    double eta = 1.0 / mu_pow(pw, xi, false);
    int it = 0, minor_total = 0, ticks = 0;
    bool restart = true, need_gh = true, done = false;
    const int hard_cap = o.max_auglag * (o.max_feval + o.max_minor + 2) + 8;
#if defined(EXA_PROFILE_PHASES)
    for (int k = 0; k < 8; k++) st.pc[k] = 0;
#endif
    while (!done && ticks < hard_cap) {
        ticks++;
        int fin = TRON_RUN;
        if (need_gh) {
            genramp_g(d, x, g);
            genramp_h(d, A);
            if (restart) {
                it++;
                st.f = genramp_f(d, x);
                st.delta = nrm2<3>(g);
                st.iter = 1;
                st.alphac = 1.0;
                st.nfev = 1;
                st.minor_iter = 1;
                restart = false;
            }
            need_gh = false;   // (a continuing solve made its NEWX test in the tick that accepted the point, see below)
        }
        if (fin == TRON_RUN) {
            tron_compute<3>(st, x, xl, xu, A, L, g, o.cgtol, 3);
            const double ftrial = genramp_f(d, x);
            st.nfev++;
            if (!(ftrial - ftrial == 0.0)) {   // non-finite inputs: give up at the last accepted point
                EXA_UNROLL
                for (int i = 0; i < 3; i++) x[i] = st.xc[i];
                it = o.max_auglag;
                fin = TRON_MAXFEVAL;
            } else if (st.nfev >= o.max_feval) {
                fin = TRON_MAXFEVAL;
            } else {
                int task;
                const bool moved = tron_evaluate<3>(st, ftrial, x, g, o.frtol, o.fatol, o.fmin, task);
                if (task != TRON_RUN) {
                    fin = task;
                } else if (moved) {
                    // NEWX test in the accepting tick (as in branch_solve): gradient at the accepted point, same bits as
                    // the evaluation at the top of the loop; a solve that ends here saves a whole trip
                    genramp_g(d, x, g);
                    st.minor_iter++;
                    if (gpnorm<3>(x, xl, xu, g) <= o.gtol) fin = TRON_CONV;
                    if (st.minor_iter >= o.max_minor) fin = TRON_MAXMINOR;
                    // (the Hessian of the ramp subproblem does not depend on x: A stays current, the next trip starts
                    //  at the model step)
                }
            }
        }
        if (fin != TRON_RUN) {
            // mpacopf_auglag_generator_kernel_cpu.jl:92-117
            minor_total += st.minor_iter;
            const double cviol = x[0] - x[1] - x[2];
            const double cnorm = fabs(cviol);
            bool terminate = false;
            if (cnorm <= eta) {
                if (cnorm <= 1e-6) {
                    terminate = true;
                } else {
                    d.mu += xi * cviol;
                    eta = eta / mu_pow(pw, xi, true);
                }
            } else {
                xi = fmin(o.mu_max, xi * 10);
                eta = 1.0 / mu_pow(pw, xi, false);
                d.xi = xi;
            }
            if (it >= o.max_auglag) terminate = true;
            if (terminate) {
                done = true;
            } else {
                restart = true;
                need_gh = true;
            }
        }
    }
    d.xi = xi;
    auglag_it = it;
    minor_it = minor_total;
}

}  // namespace exa
