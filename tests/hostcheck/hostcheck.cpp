// TEST HARNESS ONLY: compiles the product's device solver headers
// (exaadmm.jl_amd/csrc/{tron_device,branch_solver}.h) with the HOST compiler so the
// exact lane code can be compared with the oracle on CPU, before it runs on a GPU.
// Built into tests/hostcheck/libhostcheck.so by tests/test_device_code_on_host.py.
// Not part of the product; the product has no CPU execution path.
#define _GNU_SOURCE 1
#include <math.h>
#include "branch_solver.h"
#include "genramp_solver.h"

using namespace exa;

static void fill(BranchData& d, const double* param, const double* Y, double scale)
{
    for (int k = 0; k < 8; k++) {
        d.lam_[k] = param[k];
        d.rho_[k] = param[8 + k];
        d.xt_[k] = param[16 + k];
        d.Y_[k] = Y[k];
    }
    d.ls1 = param[24];
    d.ls2 = param[25];
    d.mu = param[26];
    d.scale = scale;
}

#if defined(EXA_HOST_COUNTERS)
extern "C" {
long long exa_cnt[16] = {0};   // loop counters of tron_device.h (tests/tools/count_solver_loops.py)
}
#endif

extern "C" double hostcheck_fgh(const double* x_, const double* param, const double* Y, double scale,
                                double* g_, double* Afull)
{
    BranchData d;
    fill(d, param, Y, scale);
    double x[6], g[6];
    for (int i = 0; i < 6; i++) x[i] = x_[i];
    SymMat<6> A;
    double phi[4];
    double f = branch_fgh(d, x, g, A, phi);
    for (int i = 0; i < 6; i++) {
        g_[i] = g[i];
        for (int j = 0; j < 6; j++) Afull[i + 6 * j] = A.get(i, j);
    }
    return f;
}

// the device build's sin / cos of small angles (branch_solver.h sincos_quarter_pi), n values
extern "C" void hostcheck_sincos_quarter_pi(const double* t, int n, double* sn, double* cs)
{
    for (int i = 0; i < n; i++) sincos_quarter_pi(t[i], sn[i], cs[i]);
}

extern "C" double hostcheck_f(const double* x_, const double* param, const double* Y, double scale)
{
    BranchData d;
    fill(d, param, Y, scale);
    double x[6];
    for (int i = 0; i < 6; i++) x[i] = x_[i];
    double phi[4], sc[2];
    return branch_f(d, x, phi, sc);
}

// mu^0.1 (nine = 0) / mu^0.9 (nine = 1) through the table look-up of the lane code; *from_table = 1 when the table entry was used
extern "C" double hostcheck_mu_pow(double mu, int nine, int* from_table)
{
    MuPowTable pw;
    double p = 1.0;
    for (int k = 0; k < 8; k++) {
        p *= 10.0;
        pw.pow10[k] = p;
        pw.pow01[k] = -(double)(k + 1);          // markers: a table hit is recognisable
        pw.pow09[k] = -(double)(k + 1) - 100.0;
    }
    const double r = mu_pow(pw, mu, nine != 0);
    *from_table = r < 0.0;
    return r;
}

// same contract as oracle_branch_solve
extern "C" int hostcheck_branch_solve(double* x_, const double* xl_, const double* xu_, double* param,
                                      const double* Y, double scale, int max_auglag, double mu_max,
                                      int* auglag_it, int* minor_it, int* ticks)
{
    BranchData d;
    fill(d, param, Y, scale);
    BranchOptions o;
    o.max_auglag = max_auglag;
    o.mu_max = mu_max;
    MuPowTable pw;
    double p = 1.0;
    for (int k = 0; k < 8; k++) {
        p *= 10.0;
        pw.pow10[k] = p;
        pw.pow01[k] = pow(p, 0.1);
        pw.pow09[k] = pow(p, 0.9);
    }
    double x[6], xl[6], xu[6], phi[4];
    for (int i = 0; i < 6; i++) {
        x[i] = x_[i];
        xl[i] = xl_[i];
        xu[i] = xu_[i];
    }
    BranchStats st;
    SymMat<6> A, Lm;
    branch_solve(d, o, pw, x, xl, xu, A, Lm, phi, st);
    for (int i = 0; i < 6; i++) x_[i] = x[i];
    param[24] = d.ls1;
    param[25] = d.ls2;
    param[26] = d.mu;
    *auglag_it = st.auglag_it;
    *minor_it = st.minor_it;
    *ticks = st.ticks;
    return 0;
}

// same contract as oracle_genramp_solve (scale = 1)
extern "C" int hostcheck_genramp_solve(double* x_, const double* xl_, const double* xu_, double* param, double c2,
                                       double c1, double c0, double baseMVA, int max_auglag, double xi_max,
                                       int* auglag_it, int* minor_it)
{
    GenRampData d;
    d.lam_p = param[0]; d.lam_h = param[1]; d.rho_p = param[2]; d.rho_h = param[3];
    d.xt_p = param[4]; d.xt_h = param[5]; d.mu = param[6]; d.xi = param[7];
    d.c2 = c2; d.c1 = c1; d.c0 = c0; d.M = baseMVA;
    BranchOptions o;
    o.max_auglag = max_auglag;
    o.mu_max = xi_max;
    MuPowTable pw;
    double p = 1.0;
    for (int k = 0; k < 8; k++) {
        p *= 10.0;
        pw.pow10[k] = p;
        pw.pow01[k] = pow(p, 0.1);
        pw.pow09[k] = pow(p, 0.9);
    }
    double x[3], xl[3], xu[3];
    for (int i = 0; i < 3; i++) {
        x[i] = x_[i];
        xl[i] = xl_[i];
        xu[i] = xu_[i];
    }
    genramp_solve(d, o, pw, x, xl, xu, *auglag_it, *minor_it);
    for (int i = 0; i < 3; i++) x_[i] = x[i];
    param[6] = d.mu;
    param[7] = d.xi;
    return 0;
}

// objective, gradient, Hessian of the generator ramp subproblem (lane code; scale = 1)
extern "C" double hostcheck_genramp_fgh(const double* x_, const double* param, double c2, double c1, double c0,
                                        double baseMVA, double* g_, double* Afull)
{
    GenRampData d;
    d.lam_p = param[0]; d.lam_h = param[1]; d.rho_p = param[2]; d.rho_h = param[3];
    d.xt_p = param[4]; d.xt_h = param[5]; d.mu = param[6]; d.xi = param[7];
    d.c2 = c2; d.c1 = c1; d.c0 = c0; d.M = baseMVA;
    double x[3] = {x_[0], x_[1], x_[2]}, g[3];
    SymMat<3> A;
    genramp_g(d, x, g);
    genramp_h(d, A);
    for (int i = 0; i < 3; i++) {
        g_[i] = g[i];
        for (int j = 0; j < 3; j++) Afull[i + 3 * j] = A.get(i, j);
    }
    return genramp_f(d, x);
}
