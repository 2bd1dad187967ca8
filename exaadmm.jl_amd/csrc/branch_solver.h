// Per-branch subproblem of the component-based ACOPF ADMM: objective, gradient,
// Hessian of the branch augmented Lagrangian and the augmented-Lagrangian x TRON
// loop, one branch per GPU lane.
//
// Reference behaviour reproduced (paths under /root/reference/src/models/acopf/):
//   acopf_eval_linelimit_kernel_gpu.jl:1-49     objective
//   acopf_eval_linelimit_kernel_gpu.jl:51-187   gradient
//   acopf_eval_linelimit_kernel_gpu.jl:189-594  dense 6x6 Hessian
//   acopf_auglag_linelimit_kernel_gpu.jl:76-134 augmented-Lagrangian loop
//   acopf_tron_linelimit_kernel.jl:44-144       TRON driver constants / control flow
//
// The reference expands every derivative term by term (~1500 flops per Hessian,
// computed redundantly by all 32 threads of a block).  Here the four flows are
//   phi_k = a_k vi^2 + b_k vj^2 + vi vj E_k(th),  th = ti - tj,
// with only two independent trig combinations per branch end
//   E0 = YftR cos + YftI sin,  F0 = dE0/dth,  E2 = YtfR cos - YtfI sin,  F2 = dE2/dth
// (the q-flows reuse them: E_q = -F_p, F_q = E_p), the Hessian is assembled in
// the 3 variables (vi, vj, th) from 6 weighted rank-1 terms plus the curvature
// of the flows, and expanded to (vi, vj, ti, tj) with d/dtj = -d/dti.
#pragma once
#include <cstring>
#include "tron_device.h"

namespace exa {

// Problem constants of one branch.  lam/rho/xt are rows 1-8 / 9-16 / 17-24 of the
// reference's membuf, ls1/ls2/mu rows 25/26/27 (acopf_auglag_linelimit_kernel_gpu.jl:49-79).
// Y = YffR YffI YftR YftI YttR YttI YtfR YtfI.
// BranchData owns the 32 read-only constants and the three values the augmented-Lagrangian loop
// updates; all of it lives in registers.
struct BranchData {
    double lam_[8], rho_[8], xt_[8], Y_[8];
    double ls1, ls2, mu, scale;
    EXA_HD double lam(int k) const { return lam_[k]; }
    EXA_HD double rho(int k) const { return rho_[k]; }
    EXA_HD double xt(int k) const { return xt_[k]; }
    EXA_HD double Y(int k) const { return Y_[k]; }
#if defined(__HIP_DEVICE_COMPILE__)
    // The 32 read-only constants are only consumed by the objective / derivative evaluations.  A copy is parked in LDS
    // ([entry][lane], conflict-free, 16 KB per wavefront) when the solve starts; the register copies are dead across the
    // model step (two thirds of a trip, which never looks at them) and are fetched back after it: 64 registers less at
    // the solver's pressure peak, 217 -> 140 AGPRs of spill space, a quarter of the AGPR moves gone; +2.8 % iterations/s
    // at 13 659 buses, +4 % on the whole solve (profiles/archive_r03_r04/r03_parked_constants.txt).  (Round 1's variant read the constants
    // from LDS at every use inside the evaluations and lost 3-15 %.)
    __device__ __forceinline__ void park(double* p) const
    {
        EXA_UNROLL
        for (int k = 0; k < 8; k++) {
            p[k * 64] = lam_[k];
            p[(8 + k) * 64] = rho_[k];
            p[(16 + k) * 64] = xt_[k];
            p[(24 + k) * 64] = Y_[k];
        }
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void fetch(const double* p)
    {
        asm volatile("" ::: "memory");
        EXA_UNROLL
        for (int k = 0; k < 8; k++) {
            lam_[k] = p[k * 64];
            rho_[k] = p[(8 + k) * 64];
            xt_[k] = p[(16 + k) * 64];
            Y_[k] = p[(24 + k) * 64];
        }
    }
#else
    EXA_HD void park(double*) const {}
    EXA_HD void fetch(const double*) {}
#endif
};

// Run-time parameters of the augmented-Lagrangian loop, and the TRON driver constants.  The driver
// constants are hard-wired in the reference (acopf_auglag_linelimit_kernel_gpu.jl:96,
// acopf_tron_linelimit_kernel.jl:46-51); as compile-time constants they cost no kernel-argument SGPRs
// (the kernel spills ~100 SGPRs) and fold into the comparisons.
struct BranchOptions {
    int max_auglag;    // Parameters.max_auglag (50)
    double mu_max;     // Parameters.mu_max (1e8)
    static constexpr int max_feval = 500;
    static constexpr int max_minor = 200;
    static constexpr double gtol = 1e-6;
    static constexpr double frtol = 1e-12;
    static constexpr double fatol = 0.0;
    static constexpr double fmin = -1e32;
    static constexpr double cgtol = 0.1;
};

// mu only takes the values 10^k (k = 1..8) in normal operation; host-computed mu^0.1 and
// mu^0.9 for those (same libm as the CPU path), pow() otherwise.  Kept out of the kernel
// arguments (48 SGPRs) in a small table the lanes index on demand.
struct MuPowTable {
    double pow10[8], pow01[8], pow09[8];
};

// The entry is found from mu's binary exponent -- 10^1 .. 10^8 have the exponents 3, 6, 9, 13, 16, 19, 23, 26, so
// k = (3 e + 9) / 10 -- and confirmed by comparing pow10[k - 1] with mu: one indexed read instead of a chain of eight
// comparisons.  (On the device the chain kept all 24 table values in SGPRs, which spilled: 64 v_readlane / compare pairs in
// the branch kernel.  The kernel passes a copy of the table in LDS: an indexed read there costs a lone wavefront ~100 clocks.)
#if defined(__HIP_DEVICE_COMPILE__)
// (the library pow() is ~400 instructions; inlined at three places of the solver loop it cost the lane code registers on a path that normal
//  operation never takes -- mu is a power of ten unless a caller uploaded something else.  Outlined: 30 - 46 AGPRs and ~900 instructions less
//  in every kernel that carries the branch solver, round 6)
__device__ __attribute__((noinline)) double mu_pow_generic(double mu, double e) { return pow(mu, e); }
#else
inline double mu_pow_generic(double mu, double e) { return pow(mu, e); }
#endif
EXA_HD double mu_pow(const MuPowTable& t, double mu, bool nine)
{
    long long bits;
    memcpy(&bits, &mu, sizeof(bits));
    const int e = (int)((bits >> 52) & 0x7ff) - 1023;
    int k = (3 * e + 9) / 10;
    k = k < 1 ? 1 : (k > 8 ? 8 : k);
    if (mu == t.pow10[k - 1]) return nine ? t.pow09[k - 1] : t.pow01[k - 1];
    return mu_pow_generic(mu, nine ? 0.9 : 0.1);
}

// Default (fast) device build: the compiler may fuse the multiply-add pairs of the objective / gradient / Hessian
// evaluation (252 + 234 of the kernel's 1 930 separate multiplies / adds become 248 fused ones; +2.8 % iterations/s at
// 13 659 buses).  Like the explicit fma() of the TRON algebra this rounds once where the CPU path rounds twice, so it is
// off in the exact-arithmetic build (the parity instrument).
// The rest of the library (bus update, z / l / residual pass, generators) stays at -ffp-contract=off.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EXA_EXACT_ORACLE_ARITH)
#define EXA_CONTRACT_FGH 1
#pragma clang fp contract(fast)
#endif
struct FlowTrig {
    double vi2, vj2, vv, E0, F0, E2, F2;
};

// sin / cos of the angle difference: the only transcendental of the branch model.  Evaluated once
// per point (together with the flows) and carried with the point: g and H at an accepted trial
// point reuse the values its objective evaluation produced.
// Fast device build: angle differences of a power grid are small -- for |t| <= pi/4 the two kernel polynomials of every libm
// (the minimax coefficients of fdlibm's __kernel_sin / __kernel_cos, < 1 ulp on that interval) need no argument reduction:
// ~20 multiply-adds instead of the library call (441 clocks for a lone wavefront, scripts/microbench/valu_latency.hip).
// Larger angles take the library call.
// the two kernel polynomials, valid for |t| <= pi/4 (host-callable: tests/test_device_code_on_host.py compares them with libm over
// the whole interval -- <= 1 ulp, exact at 0 -- so that a wrong coefficient is a failed CPU test, not a few extra
// iteration-count mismatches in the GPU statistics)
EXA_HD void sincos_quarter_pi(double t, double& sn, double& cs)
{
#if defined(__clang__)
#pragma clang fp contract(off)   // every instantiation of the kernels must round this the same way: the fusions are written out
#endif
    const double z = t * t;
    const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                 -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    sn = fma(z * t, fma(z, rs, -1.66666666666666324348e-01), t);
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                              2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    cs = w + fma(z, rc, (1.0 - w) - hz);
}

// EXA_FAST_SINCOS: on in the default device build, off in the exact-arithmetic build and in host builds (which may switch it on)
#if !defined(EXA_FAST_SINCOS) && defined(__HIP_DEVICE_COMPILE__) && !defined(EXA_EXACT_ORACLE_ARITH)
#define EXA_FAST_SINCOS 1
#endif
EXA_HD void angle_sincos(const double (&x)[6], double& sn, double& cs)
{
    const double t = x[2] - x[3];
#if defined(EXA_FAST_SINCOS) && EXA_FAST_SINCOS
    if (fabs(t) <= 0.78539816339744830962) {
        sincos_quarter_pi(t, sn, cs);
        return;
    }
#endif
    sincos(t, &sn, &cs);
}

template <class D>
EXA_HD void flow_trig(const D& d, const double (&x)[6], double sn, double cs, FlowTrig& t)
{
    t.vi2 = x[0] * x[0];
    t.vj2 = x[1] * x[1];
    t.vv = x[0] * x[1];
#if defined(EXA_EXACT_ORACLE_ARITH)
    t.E0 = x[0] * x[1] * cs;   // vi_vj_cos
    t.F0 = x[0] * x[1] * sn;   // vi_vj_sin
    t.E2 = t.F2 = 0.0;
    return;
#endif
    t.E0 = d.Y(2) * cs + d.Y(3) * sn;
    t.F0 = d.Y(3) * cs - d.Y(2) * sn;
    t.E2 = d.Y(6) * cs - d.Y(7) * sn;
    t.F2 = -d.Y(6) * sn - d.Y(7) * cs;
}

// flows (pij, qij, pji, qji)
template <class D>
EXA_HD void flows_from_trig(const D& d, const FlowTrig& t, double (&phi)[4])
{
#if defined(EXA_EXACT_ORACLE_ARITH)
    // operation order of the CPU path (acopf_eval_linelimit_kernel_cpu.jl:11-16, in the form the tests' CPU
    // restatement evaluates it); t.E0 / t.F0 carry vi vj cos / vi vj sin in this build
    phi[0] = d.Y(0) * t.vi2 + d.Y(2) * t.E0 + d.Y(3) * t.F0;
    phi[1] = -d.Y(1) * t.vi2 - d.Y(3) * t.E0 + d.Y(2) * t.F0;
    phi[2] = d.Y(4) * t.vj2 + d.Y(6) * t.E0 - d.Y(7) * t.F0;
    phi[3] = -d.Y(5) * t.vj2 - d.Y(7) * t.E0 - d.Y(6) * t.F0;
#else
    phi[0] = d.Y(0) * t.vi2 + t.vv * t.E0;
    phi[1] = -d.Y(1) * t.vi2 - t.vv * t.F0;
    phi[2] = d.Y(4) * t.vj2 + t.vv * t.E2;
    phi[3] = -d.Y(5) * t.vj2 + t.vv * t.F2;
#endif
}

// flows at x; sc[0] = sin, sc[1] = cos of the angle difference are returned for later g/H evaluations at x
template <class D>
EXA_HD void branch_flows(const D& d, const double (&x)[6], double (&phi)[4], double (&sc)[2])
{
    FlowTrig t;
    angle_sincos(x, sc[0], sc[1]);
    flow_trig(d, x, sc[0], sc[1], t);
    flows_from_trig(d, t, phi);
}

template <class D>
EXA_HD double objective_from_flows(const D& d, const double (&x)[6], const FlowTrig& t, const double (&phi)[4])
{
    const double y[8] = {phi[0], phi[1], phi[2], phi[3], t.vi2, t.vj2, x[2], x[3]};
    double f = 0.0;
    EXA_UNROLL
    for (int k = 0; k < 8; k++) f += d.lam(k) * y[k];
    EXA_UNROLL
    for (int k = 0; k < 8; k++) {
        double e = y[k] - d.xt(k);
        f += 0.5 * (d.rho(k) * (e * e));
    }
    const double c1 = phi[0] * phi[0] + phi[1] * phi[1] + x[4];
    const double c2 = phi[2] * phi[2] + phi[3] * phi[3] + x[5];
    f += d.ls1 * c1;
    f += d.ls2 * c2;
    f += 0.5 * (d.mu * (c1 * c1));
    f += 0.5 * (d.mu * (c2 * c2));
    return f * d.scale;
}

// objective at x; also returns the flows and sin/cos at x
template <class D>
EXA_BIG double branch_f(const D& d, const double (&x)[6], double (&phi)[4], double (&sc)[2])
{
    FlowTrig t;
    angle_sincos(x, sc[0], sc[1]);
    flow_trig(d, x, sc[0], sc[1], t);
    flows_from_trig(d, t, phi);
    return objective_from_flows(d, x, t, phi);
}

// objective at x from flows and sin/cos already known at x (same operations as branch_f after the
// transcendental, hence the same bits): used when only the multipliers / penalty changed
template <class D>
EXA_HD double branch_f_known(const D& d, const double (&x)[6], const double (&phi)[4], const double (&sc)[2])
{
    FlowTrig t;
    flow_trig(d, x, sc[0], sc[1], t);
    return objective_from_flows(d, x, t, phi);
}

// Intermediates of the gradient that the Hessian reuses (all SSA values after inlining).
struct GradParts {
    FlowTrig t;
    double E[4], F[4], J[4][3], w[4], r[4], d1[3], d2[3], wvi, wvj;
#if defined(EXA_EXACT_ORACLE_ARITH)
    double Apk[21];   // packed lower triangle of the Hessian, assembled with the gradient
#endif
};

#if defined(EXA_EXACT_ORACLE_ARITH)
// Gradient and Hessian in the operation order of the tests' CPU restatement of the reference callbacks
// (acopf_eval_linelimit_kernel_cpu.jl:48-605; chain-rule form): the flows are re-evaluated from x inside
// the derivative code and every sum runs in that restatement's order, so results are bitwise comparable.
template <class D>
EXA_BIG void branch_gh_exact(const D& d, const double (&x)[6], double (&g)[6], GradParts& q)
{
    const double a_[4] = {d.Y(0), -d.Y(1), 0.0, 0.0}, b_[4] = {0.0, 0.0, d.Y(4), -d.Y(5)};
    const double al[4] = {d.Y(2), -d.Y(3), d.Y(6), -d.Y(7)}, be[4] = {d.Y(3), d.Y(2), -d.Y(7), -d.Y(6)};
    const double vi = x[0], vj = x[1];
    const double cs = cos(x[2] - x[3]), sn = sin(x[2] - x[3]);
    const double mu = d.mu, scale = d.scale;
    double phi[4], J[4][4], H[4][4][4];
    for (int k = 0; k < 4; k++) {
        const double E = al[k] * cs + be[k] * sn;
        const double F = -al[k] * sn + be[k] * cs;
        phi[k] = a_[k] * vi * vi + b_[k] * vj * vj + vi * vj * E;
        J[k][0] = 2 * a_[k] * vi + vj * E;
        J[k][1] = 2 * b_[k] * vj + vi * E;
        J[k][2] = vi * vj * F;
        J[k][3] = -J[k][2];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) H[k][i][j] = 0.0;
        H[k][0][0] = 2 * a_[k];
        H[k][1][1] = 2 * b_[k];
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
    const double m1 = d.ls1 + mu * c1;
    const double m2 = d.ls2 + mu * c2;
    double w[4], r[4];
    for (int k = 0; k < 4; k++) {
        const double mk = (k < 2) ? m1 : m2;
        w[k] = d.lam(k) + d.rho(k) * (phi[k] - d.xt(k)) + 2 * phi[k] * mk;
        r[k] = d.rho(k) + 2 * mk;
    }
    double d1[4], d2[4];
    for (int i = 0; i < 4; i++) {
        d1[i] = 2 * (phi[0] * J[0][i] + phi[1] * J[1][i]);
        d2[i] = 2 * (phi[2] * J[2][i] + phi[3] * J[3][i]);
    }
    double gy[4];
    for (int i = 0; i < 4; i++) {
        double sacc = 0.0;
        for (int k = 0; k < 4; k++) sacc += w[k] * J[k][i];
        gy[i] = sacc;
    }
    const double wvi = d.lam(4) + d.rho(4) * (vi * vi - d.xt(4));
    const double wvj = d.lam(5) + d.rho(5) * (vj * vj - d.xt(5));
    gy[0] += wvi * (2 * vi);
    gy[1] += wvj * (2 * vj);
    gy[2] += d.lam(6) + d.rho(6) * (x[2] - d.xt(6));
    gy[3] += d.lam(7) + d.rho(7) * (x[3] - d.xt(7));
    g[0] = scale * gy[0];
    g[1] = scale * gy[1];
    g[2] = scale * gy[2];
    g[3] = scale * gy[3];
    g[4] = scale * m1;
    g[5] = scale * m2;
    double Hy[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j <= i; j++) {
            double sacc = 0.0;
            for (int k = 0; k < 4; k++) sacc += w[k] * H[k][i][j] + r[k] * J[k][i] * J[k][j];
            sacc += mu * (d1[i] * d1[j] + d2[i] * d2[j]);
            Hy[i][j] = sacc;
        }
    Hy[0][0] += 2 * wvi + 4 * d.rho(4) * vi * vi;
    Hy[1][1] += 2 * wvj + 4 * d.rho(5) * vj * vj;
    Hy[2][2] += d.rho(6);
    Hy[3][3] += d.rho(7);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j <= i; j++) q.Apk[i * (i + 1) / 2 + j] = scale * Hy[i][j];
    for (int i = 0; i < 4; i++) {
        q.Apk[4 * 5 / 2 + i] = scale * (mu * d1[i]);
        q.Apk[5 * 6 / 2 + i] = scale * (mu * d2[i]);
    }
    q.Apk[4 * 5 / 2 + 4] = scale * mu;
    q.Apk[5 * 6 / 2 + 4] = 0.0;
    q.Apk[5 * 6 / 2 + 5] = scale * mu;
}
#endif

// gradient at x from flows and sin/cos known at x
template <class D>
EXA_BIG void branch_g(const D& d, const double (&x)[6], const double (&phi)[4], const double (&scs)[2],
                      double (&g)[6], GradParts& q)
{
#if defined(EXA_EXACT_ORACLE_ARITH)
    (void)phi;
    (void)scs;
    branch_gh_exact(d, x, g, q);
    return;
#endif
    flow_trig(d, x, scs[0], scs[1], q.t);
    const FlowTrig& t = q.t;
    const double vi = x[0], vj = x[1], vv = t.vv, sc = d.scale, mu = d.mu;
    q.E[0] = t.E0; q.E[1] = -t.F0; q.E[2] = t.E2; q.E[3] = t.F2;
    q.F[0] = t.F0; q.F[1] = t.E0; q.F[2] = t.F2; q.F[3] = -t.E2;
    const double a2[4] = {2 * d.Y(0), -2 * d.Y(1), 0.0, 0.0};  // d2phi/dvi2
    const double b2[4] = {0.0, 0.0, 2 * d.Y(4), -2 * d.Y(5)};  // d2phi/dvj2

    // Jacobian of the flows in (vi, vj, th)
    EXA_UNROLL
    for (int k = 0; k < 4; k++) {
        q.J[k][0] = a2[k] * vi + vj * q.E[k];
        q.J[k][1] = b2[k] * vj + vi * q.E[k];
        q.J[k][2] = vv * q.F[k];
    }
    const double c1 = phi[0] * phi[0] + phi[1] * phi[1] + x[4];
    const double c2 = phi[2] * phi[2] + phi[3] * phi[3] + x[5];
    const double m1 = d.ls1 + mu * c1;
    const double m2 = d.ls2 + mu * c2;

    EXA_UNROLL
    for (int k = 0; k < 4; k++) {
        const double mk = (k < 2) ? m1 : m2;
        q.w[k] = d.lam(k) + d.rho(k) * (phi[k] - d.xt(k)) + 2 * phi[k] * mk;
        q.r[k] = d.rho(k) + 2 * mk;
    }
    double gr[3];
    EXA_UNROLL
    for (int i = 0; i < 3; i++) {
        q.d1[i] = 2 * (phi[0] * q.J[0][i] + phi[1] * q.J[1][i]);
        q.d2[i] = 2 * (phi[2] * q.J[2][i] + phi[3] * q.J[3][i]);
        gr[i] = q.w[0] * q.J[0][i] + q.w[1] * q.J[1][i] + q.w[2] * q.J[2][i] + q.w[3] * q.J[3][i];
    }
    q.wvi = d.lam(4) + d.rho(4) * (t.vi2 - d.xt(4));
    q.wvj = d.lam(5) + d.rho(5) * (t.vj2 - d.xt(5));

    g[0] = sc * (gr[0] + q.wvi * (2 * vi));
    g[1] = sc * (gr[1] + q.wvj * (2 * vj));
    g[2] = sc * (gr[2] + (d.lam(6) + d.rho(6) * (x[2] - d.xt(6))));
    g[3] = sc * (-gr[2] + (d.lam(7) + d.rho(7) * (x[3] - d.xt(7))));
    g[4] = sc * m1;
    g[5] = sc * m2;
}

// packed Hessian at x from the gradient's intermediates
template <class D, class MA>
EXA_BIG void branch_h(const D& d, const double (&x)[6], const GradParts& q, MA& A)
{
#if defined(EXA_EXACT_ORACLE_ARITH)
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) A.lo(i, j) = q.Apk[i * (i + 1) / 2 + j];
    return;
#endif
    const FlowTrig& t = q.t;
    const double vi = x[0], vj = x[1], vv = t.vv, sc = d.scale, mu = d.mu;
    const double* w = q.w;
    const double a2[4] = {2 * d.Y(0), -2 * d.Y(1), 0.0, 0.0};
    const double b2[4] = {0.0, 0.0, 2 * d.Y(4), -2 * d.Y(5)};
    // reduced 3x3 Hessian in (vi, vj, th): sum_k w_k Hess(phi_k) + sum_k r_k J_k J_k' + mu (d1 d1' + d2 d2')
    const double WE = w[0] * q.E[0] + w[1] * q.E[1] + w[2] * q.E[2] + w[3] * q.E[3];
    const double WF = w[0] * q.F[0] + w[1] * q.F[1] + w[2] * q.F[2] + w[3] * q.F[3];
    double Hr[3][3];
    Hr[0][0] = (w[0] * a2[0] + w[1] * a2[1]) + (2 * q.wvi + 4 * d.rho(4) * t.vi2);
    Hr[1][1] = (w[2] * b2[2] + w[3] * b2[3]) + (2 * q.wvj + 4 * d.rho(5) * t.vj2);
    Hr[1][0] = WE;
    Hr[2][0] = vj * WF;
    Hr[2][1] = vi * WF;
    Hr[2][2] = -vv * WE;
    EXA_UNROLL
    for (int i = 0; i < 3; i++) {
        EXA_UNROLL
        for (int j = 0; j <= i; j++) {
            double s = Hr[i][j];
            EXA_UNROLL
            for (int k = 0; k < 4; k++) s += (q.r[k] * q.J[k][i]) * q.J[k][j];
            s += mu * (q.d1[i] * q.d1[j] + q.d2[i] * q.d2[j]);
            Hr[i][j] = s;
        }
    }
    A.lo(0, 0) = sc * Hr[0][0];
    A.lo(1, 0) = sc * Hr[1][0];
    A.lo(1, 1) = sc * Hr[1][1];
    A.lo(2, 0) = sc * Hr[2][0];
    A.lo(2, 1) = sc * Hr[2][1];
    A.lo(3, 0) = -(sc * Hr[2][0]);
    A.lo(3, 1) = -(sc * Hr[2][1]);
    A.lo(2, 2) = sc * (Hr[2][2] + d.rho(6));
    A.lo(3, 2) = -(sc * Hr[2][2]);
    A.lo(3, 3) = sc * (Hr[2][2] + d.rho(7));
    const double smu = sc * mu;
    A.lo(4, 0) = smu * q.d1[0];
    A.lo(4, 1) = smu * q.d1[1];
    A.lo(4, 2) = smu * q.d1[2];
    A.lo(4, 3) = -(smu * q.d1[2]);
    A.lo(4, 4) = smu;
    A.lo(5, 0) = smu * q.d2[0];
    A.lo(5, 1) = smu * q.d2[1];
    A.lo(5, 2) = smu * q.d2[2];
    A.lo(5, 3) = -(smu * q.d2[2]);
    A.lo(5, 4) = 0.0;
    A.lo(5, 5) = smu;
}

// objective, gradient and packed Hessian at x in one call (tests; the solver loop uses the pieces)
template <class D, class MA>
EXA_BIG double branch_fgh(const D& d, const double (&x)[6], double (&g)[6], MA& A, double (&phi)[4])
{
    double sc[2];
    const double f = branch_f(d, x, phi, sc);
    GradParts q;
    branch_g(d, x, phi, sc, g, q);
    branch_h(d, x, q, A);
    return f;
}

#if defined(EXA_CONTRACT_FGH)
#pragma clang fp contract(off)
#endif
#if defined(EXA_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define EXA_TICK() __builtin_amdgcn_s_memtime()
#else
#define EXA_TICK() 0ull
#endif

struct BranchStats {
    unsigned long long cyc[13];  // diagnostic build only: fgh, model step, f(trial), ratio test, auglag step, then TronState::pc[0..7]
    unsigned long long* tl;     // diagnostic build only: per-wavefront timeline, (active lanes << 56 | clock) per tick
    int auglag_it;
    int minor_it;  // TRON minor iterations summed over the augmented-Lagrangian iterations
    int ticks;     // trips of the lane's solver loop
};

// Augmented-Lagrangian x TRON solve of one branch, starting from the feasible x.
// d.mu must be initialised; d.ls1, d.ls2, d.mu are updated.  phi receives the flows at
// the final x.  The two nested reference loops (auglag iterations, each a full TRON
// reverse-communication loop) are flattened into one loop of "ticks": a tick refreshes
// g (and H, if the solve goes on) when the previous tick moved x or a new subproblem starts,
// then makes one model step + ratio test.  Lanes of a wavefront therefore stay phase-aligned
// whatever auglag/TRON iteration each of them is in.
//
// Work that the reference repeats and this loop does not (same operations on the same inputs, so
// the same bits): sin/cos and the flows are evaluated once per point, by the objective evaluation
// of the trial point, and carried with it (phi/scs accepted, phit/scst trial); f is recomputed only
// when the multipliers changed (new subproblem).
// CLONES: all 64 lanes of the wavefront run THIS solve (same inputs in every lane: the heavy wavefronts of kernels.h k_resident);
// search = 2 x 6 + 2 doubles of LDS for the Cauchy search's hand-over (tron_device.h cauchy_clones).
template <bool CLONES = false, class D, class MA, class ML, class PK = NoPark>
EXA_HD void branch_solve(D& d, const BranchOptions& o, const MuPowTable& pw, double (&x)[6], double (&xl)[6],
                         double (&xu)[6], MA& A, ML& L, double (&phi)[4], BranchStats& stats, double* park = nullptr,
                         double* search = nullptr, const PK& pk = PK())
{
    d.park(park);
    pk.keep_bounds(xl, xu);
    TronState<6> st;
    double g[6];
    double scs[2];             // sin, cos of the angle difference at the accepted x
    double phit[4], scst[2];   // flows, sin/cos at the trial point

    double mu = d.mu;
    double eta = 1.0 / mu_pow(pw, mu, false);
    int it = 0, minor_total = 0, ticks = 0;
    bool restart = true, need_gh = true, done = false;
    const int hard_cap = o.max_auglag * (o.max_feval + o.max_minor + 2) + 8;

    EXA_UNROLL
    for (int k = 0; k < 13; k++) stats.cyc[k] = 0;
#if defined(EXA_PROFILE_PHASES)
    EXA_UNROLL
    for (int k = 0; k < 8; k++) st.pc[k] = 0;
#endif
    branch_flows(d, x, phi, scs);
    while (!done && ticks < hard_cap) {
        ticks++;
        int fin = TRON_RUN;
        unsigned long long t0 = EXA_TICK();
#if defined(EXA_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
        const int n_active = (int)__popcll(__ballot(1));
        if (stats.tl && ticks <= 127) stats.tl[ticks - 1] = (t0 & 0x00ffffffffffffffull) | ((unsigned long long)n_active << 56);
        // phase counters only keep ticks in which this lane ran alone (true single-lane latency)
        unsigned long long keep_c[5], keep_p[8];
        EXA_UNROLL
        for (int k = 0; k < 5; k++) keep_c[k] = stats.cyc[k];
        EXA_UNROLL
        for (int k = 0; k < 8; k++) keep_p[k] = st.pc[k];
#endif
        const bool refresh = need_gh;
        GradParts gp;
        if (refresh) {
            // ONE evaluation site for both cases (a second inlined copy would be executed one after
            // the other whenever a wavefront holds lanes of both kinds)
            branch_g(d, x, phi, scs, g, gp);
            // (the Hessian is assembled with every gradient of a new subproblem: deferring it behind the model step
            //  kept ~50 intermediates alive, +90 registers at the pressure peak, and measured 7 % slower)
            branch_h(d, x, gp, A);
            if (restart) {
                // task START: f, g, H at x; delta0 = ||g||  (tron_linelimit_kernel.jl:66-106)
                it++;
                st.f = branch_f_known(d, x, phi, scs);
                st.delta = nrm2<6>(g);
                st.iter = 1;
                st.alphac = 1.0;
                st.nfev = 1;
                st.minor_iter = 1;
                restart = false;
            }
            // (task GH of a continuing solve: its NEWX test was already made, with the same g, at the end of the tick
            //  that accepted the point -- see below; this refresh adds the Hessian)
            need_gh = false;
        }
        unsigned long long t1 = EXA_TICK();
        stats.cyc[0] += t1 - t0;
        if (fin == TRON_RUN) {
            if (PK::on) { pk.put(PK_PHI, phi); pk.put(PK_PHI + 4, scs); }
            tron_compute<6, CLONES>(st, x, xl, xu, A, L, g, o.cgtol, 6, search, pk);
            d.fetch(park);
            if (PK::on) { pk.get(PK_XC, st.xc); pk.get(PK_PHI, phi); pk.get(PK_PHI + 4, scs); }
            unsigned long long t2 = EXA_TICK();
            stats.cyc[1] += t2 - t1;
            const double ftrial = branch_f(d, x, phit, scst);
            unsigned long long t3 = EXA_TICK();
            stats.cyc[2] += t3 - t2;
            t1 = t3;
            st.nfev++;
            bool moved;
            if (!(ftrial - ftrial == 0.0)) {
                // non-finite objective (NaN / Inf reached the inputs): no comparison of the solver means
                // anything any more and the guarded loops would spin to their limits for seconds.  Give the
                // branch up at its last accepted point; the NaN stays visible in the residuals.
                EXA_UNROLL
                for (int i = 0; i < 6; i++) x[i] = st.xc[i];
                it = o.max_auglag;
                fin = TRON_MAXFEVAL;
                moved = false;
            } else if (st.nfev >= o.max_feval) {
                fin = TRON_MAXFEVAL;  // search stops before the ratio test (tron_linelimit_kernel.jl:76-78)
                moved = true;
            } else {
                int task;
                moved = tron_evaluate<6>(st, ftrial, x, g, o.frtol, o.fatol, o.fmin, task);
                if (task != TRON_RUN) {
                    fin = task;
                } else if (moved) {
                    // task GH then NEWX (tron_linelimit_kernel.jl:86-135), made HERE instead of in a tick of its own:
                    // the gradient at the accepted point from the flows / sin / cos its objective evaluation has just
                    // produced (same operations on the same inputs as the evaluation at the top of the loop, hence
                    // the same bits), then the projected-gradient test.  Nine solves in ten end on this test; they
                    // used to spend a whole extra trip on it -- gradient AND Hessian, 4.7 k clocks -- and every
                    // augmented-Lagrangian iteration of a straggler did too.  A solve that goes on assembles its
                    // Hessian here as well (below); a solve that ends does not.
                    GradParts gq;
                    branch_g(d, x, phit, scst, g, gq);
                    st.minor_iter++;
                    pk.bounds(xl, xu);
                    if (gpnorm<6>(x, xl, xu, g) <= o.gtol) fin = TRON_CONV;
                    if (st.minor_iter >= o.max_minor) fin = TRON_MAXMINOR;
                    // the solve goes on: its Hessian now, from the same intermediates -- the next trip starts at the model
                    // step with g and A current (the evaluation at the top of the loop is left to new subproblems)
                    if (fin == TRON_RUN) branch_h(d, x, gq, A);
                }
            }
            if (moved) {
                EXA_UNROLL
                for (int k = 0; k < 4; k++) phi[k] = phit[k];
                scs[0] = scst[0];
                scs[1] = scst[1];
            }
        }
        unsigned long long t4 = EXA_TICK();
        stats.cyc[3] += t4 - t1;
        if (fin != TRON_RUN) {
            // end of one TRON solve: augmented-Lagrangian update
            // (acopf_auglag_linelimit_kernel_gpu.jl:99-131)
            minor_total += st.minor_iter;
            // phi = flows at x: the same inline code evaluated them when x was accepted
            const double cviol1 = phi[0] * phi[0] + phi[1] * phi[1] + x[4];
            const double cviol2 = phi[2] * phi[2] + phi[3] * phi[3] + x[5];
            const double cnorm = fmax(fabs(cviol1), fabs(cviol2));
            bool terminate = false;
            if (cnorm <= eta) {
                if (cnorm <= 1e-6) {
                    terminate = true;
                } else {
                    d.ls1 += mu * cviol1;
                    d.ls2 += mu * cviol2;
                    eta = eta / mu_pow(pw, mu, true);
                }
            } else {
                mu = fmin(o.mu_max, mu * 10);
                eta = 1.0 / mu_pow(pw, mu, false);
                d.mu = mu;
            }
            if (it >= o.max_auglag) terminate = true;
            if (terminate) {
                done = true;
            } else {
                restart = true;
                need_gh = true;
            }
        }
        stats.cyc[4] += EXA_TICK() - t4;
#if defined(EXA_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
        if (stats.tl && ticks <= 127) stats.tl[ticks] = EXA_TICK() & 0x00ffffffffffffffull;  // end mark, overwritten by the next tick
        if (n_active != 1 && !CLONES) {   // (a heavy wavefront -- 64 clones of ONE solve -- keeps every trip: that IS its latency)
            EXA_UNROLL
            for (int k = 0; k < 5; k++) stats.cyc[k] = keep_c[k];
            EXA_UNROLL
            for (int k = 0; k < 8; k++) st.pc[k] = keep_p[k];
        } else {
            st.pc[7] += 1;
        }
#endif
    }
#if defined(EXA_PROFILE_PHASES)
    EXA_UNROLL
    for (int k = 0; k < 8; k++) stats.cyc[5 + k] = st.pc[k];
#endif
    d.mu = mu;
    stats.auglag_it = it;
    stats.minor_it = minor_total;
    stats.ticks = ticks;
}

// bounds + warm start of one branch (acopf_auglag_linelimit_kernel_gpu.jl:34-47)
// ub: current u of the branch [pij,qij,pji,qji,wi,wj,ti,tj]; bnd: (lo,hi) x (vi,vj,ti,tj); rateA: membuf row 29
EXA_HD void branch_start(const double (&ub)[8], const double (&bnd)[8], double rateA, double (&x)[6],
                         double (&xl)[6], double (&xu)[6])
{
    EXA_UNROLL
    for (int k = 0; k < 4; k++) {
        xl[k] = bnd[2 * k];
        xu[k] = bnd[2 * k + 1];
    }
    xl[4] = -rateA;
    xu[4] = 0.0;
    xl[5] = -rateA;
    xu[5] = 0.0;
    x[0] = fmin(xu[0], fmax(xl[0], sqrt(ub[4])));
    x[1] = fmin(xu[1], fmax(xl[1], sqrt(ub[5])));
    x[2] = fmin(xu[2], fmax(xl[2], ub[6]));
    x[3] = fmin(xu[3], fmax(xl[3], ub[7]));
    x[4] = fmin(xu[4], fmax(xl[4], -(ub[0] * ub[0] + ub[1] * ub[1])));
    x[5] = fmin(xu[5], fmax(xl[5], -(ub[2] * ub[2] + ub[3] * ub[3])));
}

}  // namespace exa
