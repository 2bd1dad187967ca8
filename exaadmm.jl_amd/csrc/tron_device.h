// Bound-constrained trust-region Newton solver (TRON) for one tiny dense problem
// per GPU lane, N known at compile time (N = 6 for the ACOPF branch subproblem).
//
// Replaces, for the hot path, what the reference gets from ExaTron.jl through
//   /root/reference/src/models/acopf/acopf_tron_linelimit_kernel.jl:61-144  (driver)
//   ExaTron.dtron / dcauchy / dspcg / dtrpcg / dprsrch / dicfs             (external pkg)
//
// MI355X-first design notes
//  * The reference runs ONE 32-thread block per branch and keeps x, g, A, B, L and
//    11 work vectors in shared memory, with 26-31 of 32 lanes idle.  Here one LANE
//    owns one branch and every loop has a compile-time trip count N, so after
//    unrolling all vectors and both packed-symmetric matrices live in registers while a
//    phase works on them (512 per lane on gfx950); what a phase does not touch is parked in
//    the lane's column of an LDS area by hand (LanePark below; the constants: branch_solver.h).
//    (Matrices staged in LDS and constants read from LDS at every use were measured 3-15 % slower:
//    profiles/archive_r01_r02/r02_two_waves_per_simd_experiments.txt.)
//  * The face loop of the subspace CG does not compact the free variables into
//    a smaller system (that needs run-time indexing -> scratch memory).  It keeps
//    a free-variable MASK and works on the N x N system whose bound rows/columns
//    are replaced by the identity.  In exact and in floating-point arithmetic the
//    free block sees the same operations in the same order (the extra terms are
//    exact zeros), so iterates equal those of the compacted formulation.
//  * Dense algebra (dot products, mat-vecs, Cholesky updates, triangular solves, CG updates)
//    uses explicit fma() and the triangular solves multiply by reciprocal pivots computed once per
//    factorisation: ~20 % fewer instructions and no division in the dependent chains.  Objective,
//    gradient and Hessian keep plain IEEE mul/add so that f is evaluated identically everywhere.
//    The whole GPU parity suite passes unchanged with it (per-branch iteration counts equal to the
//    CPU path on >= 99.5 % of 60 k test problems).
//  * The reverse-communication state machine of dtron is linearised by the
//    caller (branch_solver.h) into "ticks" so that the 64 lanes of a wavefront
//    stay in the same phase (model step -> f(trial) -> ratio test -> g,H).
//
// The header is also compilable by a host C++ compiler (EXA_HD empty): tests/
// build it into a CPU harness to check this exact code against the oracle.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define EXA_HD __host__ __device__ __forceinline__
#define EXA_UNROLL _Pragma("unroll")
#define EXA_BIG EXA_HD
#else
#define EXA_HD inline
#define EXA_BIG inline
#define EXA_UNROLL
#endif

#if defined(EXA_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define EXA_CLK() __builtin_amdgcn_s_memtime()
#else
#define EXA_CLK() 0ull
#endif

// -DEXA_EXACT_ORACLE_ARITH: the dense algebra is evaluated exactly as the CPU path evaluates it (the tests' CPU
// restatement of ExaTron's dense routines): products and sums rounded separately instead of fused, divisions by
// the pivots instead of multiplications by their reciprocals, the scaled / shifted Cholesky on every call, dtrpcg's
// second residual carried by recurrence, norms compared as norms.  Slower; exists so that the deviation of the
// default ("fast") arithmetic from the CPU path is a measured quantity (tests/test_gpu_exact_arith.py).
#if defined(EXA_EXACT_ORACLE_ARITH)
#define EXA_FMA(a, b, c) ((a) * (b) + (c))
#else
#define EXA_FMA(a, b, c) fma((a), (b), (c))
#endif

namespace exa {

// a / b in the TRON algebra (break points, trust-region root, CG step lengths, ratio-test curvature).  Exact build and host:
// the IEEE quotient.  Fast device build: v_rcp_f64 (relative error ~2^-23), one Newton step on the reciprocal, the quotient,
// one correction step on the quotient -- 6 dependent instructions, within an ulp of a / b for operands in the normal range --
// instead of the compiler's correctly rounded sequence (2 scalings, reciprocal, two Newton steps, quotient, correction,
// fix-up: 14 instructions, 78 clocks for a lone wavefront: scripts/microbench/valu_latency.hip).  A solver trip makes ~25
// divisions.  b = 0 or non-finite gives NaN where IEEE gives an infinity or zero: every call site discards such a quotient
// by selection (see there).
EXA_HD double exa_div(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EXA_EXACT_ORACLE_ARITH)
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
#else
    return a / b;
#endif
}

// task / termination codes (numbering of acopf_tron_linelimit_kernel.jl)
enum { TRON_RUN = 0, TRON_CONV = 4, TRON_WARN = 10, TRON_MAXMINOR = 20, TRON_MAXFEVAL = 21 };

// Packed lower triangle of a symmetric N x N matrix, (i,j) with i >= j at i(i+1)/2 + j.
// SymMat owns its storage (registers after unrolling).
// Host-only loop counters for tests/tools/count_solver_loops.py (how many trips do the searches of a model step make?).
#if defined(EXA_HOST_COUNTERS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" long long exa_cnt[16];
#define EXA_CNT(i) (exa_cnt[i]++)
#else
#define EXA_CNT(i) ((void)0)
#endif
enum { CNT_CAUCHY = 0, CNT_CAUCHY_INTERP, CNT_CAUCHY_TRIP_IN, CNT_CAUCHY_TRIP_OUT, CNT_PRSRCH, CNT_PRSRCH_TRIP, CNT_FACE, CNT_CG,
       CNT_ICFS_SLOW, CNT_CAUCHY_FIRST_OUT };

template <int N>
struct SymMat {
    double a[N * (N + 1) / 2];
    EXA_HD double get(int i, int j) const { return i >= j ? a[i * (i + 1) / 2 + j] : a[j * (j + 1) / 2 + i]; }
    EXA_HD double& lo(int i, int j) { return a[i * (i + 1) / 2 + j]; }  // requires i >= j
    EXA_HD double lo(int i, int j) const { return a[i * (i + 1) / 2 + j]; }
};

template <int N>
EXA_HD double dot(const double (&a)[N], const double (&b)[N])
{
    double s = 0.0;
    EXA_UNROLL
    for (int i = 0; i < N; i++) s = EXA_FMA(a[i], b[i], s);
    return s;
}

template <int N>
EXA_HD double nrm2(const double (&a)[N])
{
    return sqrt(dot<N>(a, a));
}

// w = A x (A symmetric, packed)
template <int N, class MA>
EXA_HD void symv(const MA& A, const double (&x)[N], double (&w)[N])
{
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        double s = 0.0;
        EXA_UNROLL
        for (int j = 0; j < N; j++) s = EXA_FMA(A.get(i, j), x[j], s);
        w[i] = s;
    }
}

// projected gradient inf-norm (ExaTron.dgpnorm; call site tron_linelimit_kernel.jl:122)
template <int N>
EXA_HD double gpnorm(const double (&x)[N], const double (&xl)[N], const double (&xu)[N], const double (&g)[N])
{
    double r = 0.0;
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        const double at_lo = fabs(fmin(g[i], 0.0)), at_hi = fabs(fmax(g[i], 0.0));
        double v = fabs(g[i]);
        v = (x[i] == xu[i]) ? at_hi : v;
        v = (x[i] == xl[i]) ? at_lo : v;
        r = (xl[i] != xu[i]) ? fmax(r, v) : r;
    }
    return r;
}

// ||a|| > delta, decided exactly as the comparison of the (correctly rounded) norm with delta decides it, but without the
// square root wherever the squares are not within rounding distance of each other: 1e-14 is ~90 ulps, far outside what
// the roundings of delta^2 and of the root can move; only near-ties -- delta is often exactly the norm of an earlier step
// -- take the root (a lone wavefront pays 103 clocks for it, the Cauchy search asks this question ~4 times per model step).
template <int N>
EXA_HD bool norm_exceeds(const double (&a)[N], double delta)
{
#if defined(EXA_EXACT_ORACLE_ARITH)
    return nrm2<N>(a) > delta;
#else
    const double n2 = dot<N>(a, a), d2 = delta * delta;
    const bool far_out = n2 > d2 * (1.0 + 1e-14), far_in = n2 < d2 * (1.0 - 1e-14);
    bool out = far_out;
    if (!(far_out || far_in)) out = sqrt(n2) > delta;
    return out;
#endif
}

// Smallest and largest break point of the ray x + alpha w in the box (TRON dbreakpt); none -> 0, 0.
// dl = xl - x, du = xu - x from the caller (x < xu exactly when du > 0).  Written without branches: every quotient is
// formed (a division by w = 0 yields a value no selection ever takes), so the six divisions overlap instead of sitting one
// after the other in six divergent regions -- ~330 clocks per call for a lone wavefront, same operands, same bits.
template <int N>
EXA_HD void breakpt(const double (&dl)[N], const double (&du)[N], const double (&w)[N], double& brptmin, double& brptmax)
{
    bool any = false;
    double bmin = 0.0, bmax = 0.0;
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        const bool up = (du[i] > 0.0) && (w[i] > 0.0);
        const bool dn = (dl[i] < 0.0) && (w[i] < 0.0);
        const bool mv = up || dn;
        const double brpt = exa_div(up ? du[i] : dl[i], w[i]);   // (w = 0: mv is false, the quotient is never selected)
        const double lo = any ? fmin(brpt, bmin) : brpt;
        const double hi = any ? fmax(brpt, bmax) : brpt;
        bmin = mv ? lo : bmin;
        bmax = mv ? hi : bmax;
        any = any || mv;
    }
    brptmin = bmin;
    brptmax = bmax;
}

// s = P[x + alpha w] - x
template <int N>
EXA_HD void gpstep(const double (&x)[N], const double (&xl)[N], const double (&xu)[N], double alpha,
                   const double (&w)[N], double (&s)[N])
{
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        double aw = alpha * w[i];
        double t = x[i] + aw;
        s[i] = (t < xl[i]) ? (xl[i] - x[i]) : ((t > xu[i]) ? (xu[i] - x[i]) : aw);
    }
}

// The same step with the distances to the bounds dl = xl - x, du = xu - x computed by the caller: the searches project
// from ONE point along ONE direction for several alpha, and with both arms of the selection ready-made the compiler emits
// selects where the form above gets two nested divergent branches per component (same values, same bits).
template <int N>
EXA_HD void gpstep_d(const double (&x)[N], const double (&xl)[N], const double (&xu)[N], const double (&dl)[N],
                     const double (&du)[N], double alpha, const double (&w)[N], double (&s)[N])
{
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        const double aw = alpha * w[i];
#if defined(EXA_EXACT_ORACLE_ARITH) || !defined(__HIP_DEVICE_COMPILE__)
        const double t = x[i] + aw;
        const double hi = (t > xu[i]) ? du[i] : aw;
        s[i] = (t < xl[i]) ? dl[i] : hi;
#else
        // fast device build: the same projection as a clamp of the step instead of a test of the point -- two instructions
        // (v_min_f64, v_max_f64) where the form above needs an add, two compares and four 32-bit selects per component, ~6
        // projections per model step.  The two forms pick different arms only when x + alpha w lands within a rounding
        // error of a bound it does not sit on; the step then differs by an ulp of x (the point itself is clamped
        // separately).  A variable that sits ON its bound (dl or du exactly 0) takes the same arm in both.
        (void)x; (void)xl; (void)xu;
        s[i] = fmax(dl[i], fmin(aw, du[i]));
#endif
    }
}

// q(s) = 1/2 s'As + g's and g's; wa receives A s
template <int N, class MA>
EXA_HD void quad_model(const MA& A, const double (&g)[N], const double (&s)[N], double& q, double& gts, double (&wa)[N])
{
    symv<N>(A, s, wa);
    gts = dot<N>(g, s);
    q = EXA_FMA(0.5, dot<N>(s, wa), gts);
}

// Cauchy step (TRON dcauchy).  alpha persists across calls.
// The break points are only needed by the extrapolation branch and are computed there.  (||s|| must be
// compared with delta as a norm, not as squares: delta is often exactly the norm of the previous step,
// so ties are structural and a squared comparison flips them.)
// as: A s of the RETURNED step when the last model evaluation of the search was made at exactly that step (the
// interpolation branch ends that way), as_valid says so: dspcg starts with the same product and need not repeat it.
template <int N, class MA>
EXA_BIG void cauchy(const double (&x)[N], const double (&xl)[N], const double (&xu)[N], const MA& A,
                   const double (&g)[N], double delta, double& alpha, double (&s)[N], double (&as)[N], bool& as_valid)
{
    const double mu0 = 0.01, interpf = 0.1, extrapf = 10.0;

    bool interp;
    as_valid = false;
    EXA_CNT(CNT_CAUCHY);
    double dl[N], du[N];
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        dl[i] = xl[i] - x[i];
        du[i] = xu[i] - x[i];
    }
    gpstep_d<N>(x, xl, xu, dl, du, -alpha, g, s);
    if (norm_exceeds<N>(s, delta)) {
        interp = true;
        EXA_CNT(CNT_CAUCHY_FIRST_OUT);
    } else {
        double q, gts;
        quad_model<N>(A, g, s, q, gts, as);
        interp = (q >= mu0 * gts);
    }

    // ONE search loop for both kinds of lanes (a wavefront that holds interpolating and extrapolating lanes used to run
    // the two loops one after the other): per lane the operations and their order are those of dcauchy's two loops.
    //   interpolation : alpha shrinks 10x per trip until the step is inside the region with sufficient decrease
    //   extrapolation : alpha grows 10x per trip while the step stays inside the region (up to the largest break
    //                   point); the last alpha that gave sufficient decrease is kept
    double brptmax = 0.0;
    if (interp) EXA_CNT(CNT_CAUCHY_INTERP);
    if (!interp) {
        double mg[N];
        EXA_UNROLL
        for (int i = 0; i < N; i++) mg[i] = -g[i];
        double brptmin;
        breakpt<N>(dl, du, mg, brptmin, brptmax);
    }
    double alphas = alpha;
    bool search = true;
    // 400 trips take any finite alpha to 0 / past every break point (guard against NaN spins)
    for (int guard = 0; search && (interp || alpha <= brptmax) && guard < 400; guard++) {
        alpha = (interp ? interpf : extrapf) * alpha;
        gpstep_d<N>(x, xl, xu, dl, du, -alpha, g, s);
        const bool inside = !norm_exceeds<N>(s, delta);
        double q = 0.0, gts = 0.0;
        if (inside) quad_model<N>(A, g, s, q, gts, as);   // (evaluating the model unconditionally measured 4 % slower: profiles/archive_r03_r04/r03_branch_solver_steps.txt)
        if (inside) EXA_CNT(CNT_CAUCHY_TRIP_IN); else EXA_CNT(CNT_CAUCHY_TRIP_OUT);
        if (interp) {
            as_valid = inside;
            if (inside) search = (q > mu0 * gts);
        } else if (inside) {
            if (q < mu0 * gts) alphas = alpha;
        } else {
            search = false;
        }
    }
    if (!interp) {
        alpha = alphas;
        gpstep_d<N>(x, xl, xu, dl, du, -alpha, g, s);
        as_valid = false;
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- the Cauchy search of a branch whose solve runs on all 64 lanes of a wavefront as clones (kernels.h: the heavy wavefronts of k_resident)
// The step lengths the search tries are known up front: alpha, then 0.1^k alpha (interpolation) or 10^k alpha (extrapolation); a
// straggler makes 3.8 model evaluations per search one after the other (profiles/archive_r01_r02/r02_solver_loop_trip_counts.txt).  Every lane holds
// the same x, bounds, g, A, delta, alpha; lane 0 evaluates alpha, lanes 1..C the interpolation chain, lanes C+1..2C the extrapolation
// chain -- gpstep_d / norm_exceeds / quad_model as the serial search calls them, on the operands the serial trip would see, so
// every candidate carries the serial trip's bits -- and wavefront ballots replay the serial loop's decisions (first sufficient
// decrease inside the region; last sufficient decrease before the step leaves the region or passes the last break point).  If a chain
// of C candidates does not end the search the pass is repeated from its last alpha (C divides the serial loop's guard of 400 trips:
// the guard falls on a pass boundary).  The winner's alpha, s, A s reach the other lanes through LDS (sh: 2 N + 2 doubles).
#ifndef EXA_COOP_CHAIN
#define EXA_COOP_CHAIN 8
#endif
__device__ __forceinline__ double lane_value(double v, int lane)   // lane: wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int N, class MA>
__device__ __forceinline__ void cauchy_clones(double* sh, const double (&x)[N], const double (&xl)[N], const double (&xu)[N], const MA& A,
                                              const double (&g)[N], double delta, double& alpha, double (&s)[N], double (&as)[N],
                                              bool& as_valid)
{
    constexpr int C = EXA_COOP_CHAIN;
    static_assert(400 % C == 0 && 2 * C + 1 <= 64 && C < 32, "chain length");
    const double mu0 = 0.01, interpf = 0.1, extrapf = 10.0;
    const int lane = (int)(threadIdx.x & 63);
    double dl[N], du[N];
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        dl[i] = xl[i] - x[i];
        du[i] = xu[i] - x[i];
    }
    const int k = (lane == 0 || lane > 2 * C) ? 0 : (lane <= C ? lane : lane - C);   // the lane's position in its chain
    const double f = (lane > C) ? extrapf : interpf;
    constexpr unsigned long long CM = (1ull << C) - 1;
    bool interp = false;
    double base = alpha, alphas = alpha, brptmax = 0.0;
    for (int pass = 0;; pass++) {   // (wave-uniform)
        double a = base, ap = base;   // the candidate and the one before it in the chain
        EXA_UNROLL
        for (int j = 1; j <= C; j++) {
            const bool on = (j <= k);
            ap = on ? a : ap;
            a = on ? f * a : a;
        }
        gpstep_d<N>(x, xl, xu, dl, du, -a, g, s);
        const bool ins = !norm_exceeds<N>(s, delta);
        double q, gts;
        quad_model<N>(A, g, s, q, gts, as);
        const double thr = mu0 * gts;
        const unsigned long long m_in = __ballot(ins), m_gt = __ballot(q > thr);
        if (pass == 0) {
            const unsigned long long m_ge = __ballot(q >= thr);
            interp = !(m_in & 1ull) || (m_ge & 1ull);
            if (!interp) {
                double mg[N];
                EXA_UNROLL
                for (int i = 0; i < N; i++) mg[i] = -g[i];
                double brptmin;
                breakpt<N>(dl, du, mg, brptmin, brptmax);
            }
        }
        const bool guard_end = (pass + 1) * C >= 400;
        if (interp) {
            const unsigned stop = (unsigned)(((m_in & ~m_gt) >> 1) & CM);   // inside the region and q <= mu0 g's
            if (stop != 0u || guard_end) {
                const int w = stop ? (__builtin_ctz(stop) + 1) : C;
                if (lane == w) {
                    EXA_UNROLL
                    for (int i = 0; i < N; i++) {
                        sh[i] = s[i];
                        sh[N + i] = as[i];
                    }
                    sh[2 * N] = a;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                EXA_UNROLL
                for (int i = 0; i < N; i++) {
                    s[i] = sh[i];
                    as[i] = sh[N + i];
                }
                alpha = sh[2 * N];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                as_valid = ((m_in >> w) & 1ull) != 0ull;
                return;
            }
            base = lane_value(a, C);
        } else {
            const unsigned long long m_lt = __ballot(q < thr), m_val = __ballot(ap <= brptmax);
            // trip j is made while the search is on and alpha_(j-1) <= brptmax; it ends the search when its step is outside
            const unsigned end = (unsigned)((~(m_val & m_in) >> (C + 1)) & CM);
            const int K = end ? (__builtin_ctz(end) + 1) : C + 1;
            const unsigned lt = (unsigned)(((m_in & m_lt) >> (C + 1)) & ((1ull << (K - 1)) - 1ull));
            if (lt != 0u) alphas = lane_value(a, C + (32 - __builtin_clz(lt)));
            if (end != 0u || guard_end) {
                alpha = alphas;
                gpstep_d<N>(x, xl, xu, dl, du, -alpha, g, s);
                as_valid = false;
                return;
            }
            base = lane_value(a, 2 * C);
        }
    }
}
#endif

template <int N, class ML>
EXA_HD void trsol_n(const ML& L, const double (&id)[N], double (&r)[N])
{
    EXA_UNROLL
    for (int j = 0; j < N; j++) {
#if defined(EXA_EXACT_ORACLE_ARITH)
        double t = r[j] / L.lo(j, j);
#else
        double t = r[j] * id[j];
#endif
        EXA_UNROLL
        for (int k = j + 1; k < N; k++) r[k] = EXA_FMA(-L.lo(k, j), t, r[k]);
        r[j] = t;
    }
}
template <int N, class ML>
EXA_HD void trsol_t(const ML& L, const double (&id)[N], double (&r)[N])
{
#if defined(EXA_EXACT_ORACLE_ARITH)
    r[N - 1] = r[N - 1] / L.lo(N - 1, N - 1);
#else
    r[N - 1] = r[N - 1] * id[N - 1];
#endif
    EXA_UNROLL
    for (int j = N - 2; j >= 0; j--) {
        double t = 0.0;
        EXA_UNROLL
        for (int k = j + 1; k < N; k++) t = EXA_FMA(L.lo(k, j), r[k], t);
#if defined(EXA_EXACT_ORACLE_ARITH)
        r[j] = (r[j] - t) / L.lo(j, j);
#else
        r[j] = (r[j] - t) * id[j];
#endif
    }
}

// sqrt(x) and 1 / sqrt(x) from ONE Goldschmidt iteration (fast device build).  The compiler's double-precision square root
// is v_rsq_f64 refined by exactly this coupled iteration, which carries h ~ 1 / (2 sqrt x) along for free; the pivot's
// reciprocal is 2h after one correction step instead of a full IEEE division (13 dependent instructions, 78 clocks for a
// lone wavefront, next to 103 for the square root: scripts/microbench/valu_latency.hip).  s is the correctly rounded
// root for normal x > 0 (the same operations the compiler emits, minus its range scaling for x < 2^-767); r is within an
// ulp of 1 / s.  x <= 0 or NaN yields NaN / garbage: the caller tests the pivot's sign itself.
EXA_HD void sqrt_rcp(double x, double& s, double& r)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EXA_EXACT_ORACLE_ARITH)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    s = g;
    const double r0 = h + h;
    r = fma(fma(-g, r0, 1.0), r0, r0);
#else
    s = sqrt(x);
    r = 1.0 / s;
#endif
}

// In-place left-looking Cholesky of the packed lower triangle (TRON dicf, dense), columns in order.
// rd[j] receives 1 / L(j,j).  Returns false when a pivot is not positive (the factor is then garbage).
template <int N, class ML>
EXA_HD bool chol_inplace(ML& L, double (&rd)[N])
{
    bool ok = true;
    EXA_UNROLL
    for (int j = 0; j < N; j++) {
        EXA_UNROLL
        for (int k = 0; k < j; k++) {
            double ljk = L.lo(j, k);
            EXA_UNROLL
            for (int i = j; i < N; i++) L.lo(i, j) = EXA_FMA(-L.lo(i, k), ljk, L.lo(i, j));
        }
        double d = L.lo(j, j), r;
        ok = ok && (d > 0.0);
        sqrt_rcp(d, d, r);
        L.lo(j, j) = d;
        rd[j] = r;
        EXA_UNROLL
#if defined(EXA_EXACT_ORACLE_ARITH)
        for (int i = j + 1; i < N; i++) L.lo(i, j) /= d;
#else
        for (int i = j + 1; i < N; i++) L.lo(i, j) *= r;
#endif
    }
    return ok;
}

// Scaled / shifted Cholesky of the free block of A (TRON dicfs + dicf, dense).
// fr: bit i set = variable i free.  Bound rows/cols of L are the identity.  id[i] = 1 / L(i,i).
//
// dicfs factorises D A D + alpha I with D = diag(||A e_j||^-1/2) and returns D^-1 L.  The scaling only
// matters for the size of the shift alpha; when no shift is needed (all free diagonals positive and
// the matrix positive definite -- >99.9 % of the calls on the ACOPF path) D^-1 chol(D A D) = chol(A)
// up to rounding, so the lane code first tries the plain factorisation of the masked matrix and only
// falls back to the scaled / shifted loop when a pivot fails.  This skips 12 square roots, 6 divisions
// and ~140 multiplications per call; the factor differs from the scaled one in the last bits only
// (it preconditions CG; parity suite unchanged).
template <int N, class MA, class ML>
EXA_BIG void icfs_masked(unsigned fr, const MA& A, ML& L, double (&id)[N])
{
    const int nbmax = 3;
    const double alpham = 1.0e-3, nbfactor = 512.0;

    double m[N];  // 1.0 on free variables, 0.0 on bound ones
    bool pos = true;
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        const bool f = (fr >> i) & 1u;
        m[i] = f ? 1.0 : 0.0;
        pos = pos && (!f || A.lo(i, i) > 0.0);
    }
#if defined(EXA_EXACT_ORACLE_ARITH)
    pos = false;   // always the scaled / shifted factorisation of dicfs
#endif
    if (pos) {
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            EXA_UNROLL
            for (int j = 0; j < i; j++) L.lo(i, j) = (A.lo(i, j) * m[i]) * m[j];
            L.lo(i, i) = ((fr >> i) & 1u) ? A.lo(i, i) : 1.0;
        }
        if (chol_inplace<N>(L, id)) return;
    }
    EXA_CNT(CNT_ICFS_SLOW);

    double wa2[N], wi2[N], rdv[N];
    EXA_UNROLL
    for (int j = 0; j < N; j++) {
        double s = 0.0;
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            double aij = A.get(i, j) * m[i];
            s = EXA_FMA(aij, aij, s);
        }
        double nrm = sqrt(s);
        const bool sc = ((fr >> j) & 1u) && nrm > 0.0;
        const double rt = sqrt(nrm);
        wa2[j] = sc ? 1.0 / rt : 1.0;
        wi2[j] = sc ? rt : 1.0;  // 1 / wa2[j] up to rounding: the final un-scaling multiplies instead of dividing
    }

    double alphas = alpham;
    double alpha = 0.0;
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        if ((fr >> i) & 1u) {
            double aii = A.lo(i, i);
            alpha = (aii == 0.0) ? alphas : fmax(alpha, -aii * (wa2[i] * wa2[i]));
        }
    }
    if (alpha > 0.0) alpha = fmax(alpha, alphas);

    int nb = 1;
    for (int guard = 0; guard < 200; guard++) {
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            EXA_UNROLL
            for (int j = 0; j <= i; j++) {
                bool f2 = ((fr >> i) & 1u) && ((fr >> j) & 1u);
                double v;
                if (i == j)
                    v = f2 ? A.lo(i, i) * (wa2[i] * wa2[i]) + alpha : 1.0;
                else
                    v = f2 ? A.lo(i, j) * wa2[i] * wa2[j] : 0.0;
                L.lo(i, j) = v;
            }
        }
        const bool ok = chol_inplace<N>(L, rdv);
        if (ok) {
            if (alpha == alphas && nb < nbmax) {
                alphas = alphas / nbfactor;
                alpha = alphas;
                nb++;
            } else {
                // L <- D^-1 L with D = diag(wa2); 1/L(i,i) = wa2[i] / d_i
                EXA_UNROLL
                for (int i = 0; i < N; i++) {
                    EXA_UNROLL
#if defined(EXA_EXACT_ORACLE_ARITH)
                    for (int j = 0; j <= i; j++) L.lo(i, j) = L.lo(i, j) / wa2[i];
#else
                    for (int j = 0; j <= i; j++) L.lo(i, j) = L.lo(i, j) * wi2[i];
#endif
                    id[i] = rdv[i] * wa2[i];
                }
                return;
            }
        } else {
            alpha = fmax(2.0 * alpha, alphas);
        }
    }
    EXA_UNROLL
    for (int i = 0; i < N; i++) id[i] = 1.0 / L.lo(i, i);
}

template <int N>
EXA_HD double trqsol(const double (&x)[N], const double (&p)[N], double delta)
{
    double ptx = dot<N>(p, x), ptp = dot<N>(p, p), xtx = dot<N>(x, x);
    double dsq = delta * delta;
    double rad = ptx * ptx + ptp * (dsq - xtx);
    rad = sqrt(fmax(rad, 0.0));
    // ONE division on selected operands instead of one division in each of two divergent branches (same operands, same
    // quotient): a branch region costs a lone wavefront ~33 clocks of VALU -> SALU -> VALU hand-over before it does anything
    const bool pos = (ptx > 0.0);
    const double q = exa_div(pos ? (dsq - xtx) : (rad - ptx), pos ? (ptx + rad) : ptp);   // (p = 0 and ptx <= 0: rad = 0, q unused)
    return (pos || rad > 0.0) ? q : 0.0;
}

// mask helper: zero the bound entries
template <int N>
EXA_HD void mask_vec(unsigned fr, double (&v)[N])
{
    EXA_UNROLL
    for (int i = 0; i < N; i++) v[i] = ((fr >> i) & 1u) ? v[i] : 0.0;
}

// Preconditioned CG on the free block inside || L' w || <= delta  (TRON dtrpcg).
// gf, w are zero on bound entries.  Returns info (1,2 converged; 3 neg. curvature; 4 boundary; 5 itermax).
// dtrpcg carries the residual twice: r in the preconditioned variables and t = L r in the original
// ones (t only feeds the test ||t|| <= tol).  Here t is not stored: ||L r|| is evaluated from r when
// the test is made (21 FMAs) -- two work vectors (t and the copy of A L^-T p that updates it) less at
// the kernel's register-pressure peak; the tested value differs from the recurrence in the last bits.
template <int N, class MA, class ML>
EXA_BIG int trpcg_masked(unsigned fr, const MA& A, const double (&gf)[N], double delta, const ML& L,
                        const double (&id)[N], double tol, double stol, int itermax, double (&w)[N], int& iters_out)
{
    double p[N], q[N], r[N], z[N];
#if defined(EXA_EXACT_ORACLE_ARITH)
    double t[N], zc[N];
#endif
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        w[i] = 0.0;
        r[i] = -gf[i];
#if defined(EXA_EXACT_ORACLE_ARITH)
        t[i] = r[i];
#endif
    }
    trsol_n<N>(L, id, r);
    EXA_UNROLL
    for (int i = 0; i < N; i++) p[i] = r[i];

    double rho = dot<N>(r, r);
    if (rho == 0.0) {
        iters_out = 0;
        return 1;
    }
    const double tol2 = tol * tol, stol2 = stol * stol;   // norms are only compared: use squares

    for (int iters = 1; iters <= itermax; iters++) {
        EXA_CNT(CNT_CG);
        EXA_UNROLL
        for (int i = 0; i < N; i++) z[i] = p[i];
        trsol_t<N>(L, id, z);
        symv<N>(A, z, q);
        mask_vec<N>(fr, q);
#if defined(EXA_EXACT_ORACLE_ARITH)
        EXA_UNROLL
        for (int i = 0; i < N; i++) zc[i] = q[i];
#endif
        trsol_n<N>(L, id, q);

        double ptq = dot<N>(p, q);
        const double rq = exa_div(rho, ptq);            // (formed outside any branch, see trqsol; unused when ptq <= 0)
        double alpha = (ptq > 0.0) ? rq : 0.0;
        double sigma = trqsol<N>(w, p, delta);

        if (ptq <= 0.0 || alpha >= sigma) {
            EXA_UNROLL
            for (int i = 0; i < N; i++) w[i] = EXA_FMA(sigma, p[i], w[i]);
            iters_out = iters;
            return (ptq <= 0.0) ? 3 : 4;
        }
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            w[i] = EXA_FMA(alpha, p[i], w[i]);
            r[i] = EXA_FMA(-alpha, q[i], r[i]);
        }
        double rtr = dot<N>(r, r);
#if defined(EXA_EXACT_ORACLE_ARITH)
        EXA_UNROLL
        for (int i = 0; i < N; i++) t[i] = EXA_FMA(-alpha, zc[i], t[i]);
        (void)tol2;
        (void)stol2;
        if (nrm2<N>(t) <= tol) {
            iters_out = iters;
            return 1;
        }
        if (sqrt(rtr) <= stol) {
            iters_out = iters;
            return 2;
        }
#else
        // ||t||^2 with t = L r (lower triangular product, row by row)
        double tt = 0.0;
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            double ti = 0.0;
            EXA_UNROLL
            for (int j = 0; j <= i; j++) ti = EXA_FMA(L.lo(i, j), r[j], ti);
            tt = EXA_FMA(ti, ti, tt);
        }
        if (tt <= tol2) {
            iters_out = iters;
            return 1;
        }
        if (rtr <= stol2) {
            iters_out = iters;
            return 2;
        }
#endif
        double beta = exa_div(rtr, rho);   // (rho > stol2 >= 0 here: the loop has returned otherwise)
        EXA_UNROLL
        for (int i = 0; i < N; i++) p[i] = EXA_FMA(beta, p[i], r[i]);
        rho = rtr;
    }
    iters_out = itermax;
    return 5;
}

// Projected search along w from x on the free block (TRON dprsrch); w <- actual step.
template <int N, class MA>
EXA_BIG void prsrch_masked(unsigned fr, double (&x)[N], const double (&xl)[N], const double (&xu)[N],
                          const MA& A, const double (&gf)[N], double (&w)[N])
{
    const double mu0 = 0.01, interpf = 0.5;
    double wa1[N], wa2[N];
    double alpha = 1.0;
    double dl[N], du[N];
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        dl[i] = xl[i] - x[i];
        du[i] = xu[i] - x[i];
    }
    double brptmin, brptmax;
    breakpt<N>(dl, du, w, brptmin, brptmax);
    EXA_CNT(CNT_PRSRCH);

    bool search = true;
    bool wa1_current = false;   // wa1 = P[x + alpha w] - x for the CURRENT alpha (the accepted trial's step need not be projected twice)
    for (int guard = 0; search && alpha > brptmin && guard < 1100; guard++) {
        EXA_CNT(CNT_PRSRCH_TRIP);
        gpstep_d<N>(x, xl, xu, dl, du, alpha, w, wa1);
        symv<N>(A, wa1, wa2);
        mask_vec<N>(fr, wa2);
        double gts = dot<N>(gf, wa1);
        double q = EXA_FMA(0.5, dot<N>(wa1, wa2), gts);
        if (q <= mu0 * gts) {
            search = false;
            wa1_current = true;
        } else {
            alpha = interpf * alpha;
        }
    }
    if (alpha < 1.0 && alpha < brptmin) {
        alpha = brptmin;
        wa1_current = false;
    }

    if (!wa1_current) gpstep_d<N>(x, xl, xu, dl, du, alpha, w, wa1);
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        double xi = x[i] + alpha * w[i];
        x[i] = fmax(xl[i], fmin(xi, xu[i]));
        w[i] = wa1[i];
    }
}

// Hand placement of solver state (round 6; first built for the two-wavefronts-per-SIMD experiment, profiles/r06_two_waves_per_simd.txt, kept because it
// pays at one): spcg hands the vectors its CG does not touch to a "parker" in front of the factorisation / CG of a face and asks for them back behind
// it.  NoPark: nothing happens (host builds, the 3-variable ramp solver).  LanePark: the lane's own column of an LDS area, [slot][lane]
// (conflict-free 512-byte rows).  Data movement only: every arithmetic operation and its order are those of the unparked code.
struct NoPark {
    template <int N> EXA_HD void put(int, const double (&)[N]) const {}
    template <int N> EXA_HD void get(int, double (&)[N]) const {}
    template <int N> EXA_HD void bounds(double (&)[N], double (&)[N]) const {}       // (the caller's arrays hold them)
    template <int N> EXA_HD void keep_bounds(const double (&)[N], const double (&)[N]) const {}
    static constexpr bool on = false;
};
// slots of the parked state
enum { PK_X = 0, PK_G = 6, PK_S = 12, PK_XC = 18, PK_BND = 24, PK_PHI = 33, PK_SLOTS = 39 };
#if defined(__HIP_DEVICE_COMPILE__)
struct LanePark {
    double* p;   // this lane's column: slot k at p[64 * k]
    template <int N> __device__ __forceinline__ void put(int slot, const double (&v)[N]) const
    {
        EXA_UNROLL
        for (int i = 0; i < N; i++) p[(slot + i) * 64] = v[i];
        asm volatile("" ::: "memory");
    }
    template <int N> __device__ __forceinline__ void get(int slot, double (&v)[N]) const
    {
        asm volatile("" ::: "memory");
        EXA_UNROLL
        for (int i = 0; i < N; i++) v[i] = p[(slot + i) * 64];
    }
    // The box of the branch subproblem never changes during a solve: it LIVES in LDS (9 values: four (lo, hi) pairs and -rateA; the upper
    // bound of the two slack variables is 0) and is read into registers in front of every stretch of code that uses it, so that it occupies no
    // registers in between.
    __device__ __forceinline__ void keep_bounds(const double (&xl)[6], const double (&xu)[6]) const
    {
        EXA_UNROLL
        for (int i = 0; i < 4; i++) {
            p[(PK_BND + 2 * i) * 64] = xl[i];
            p[(PK_BND + 2 * i + 1) * 64] = xu[i];
        }
        p[(PK_BND + 8) * 64] = xl[4];
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void bounds(double (&xl)[6], double (&xu)[6]) const
    {
        asm volatile("" ::: "memory");
        EXA_UNROLL
        for (int i = 0; i < 4; i++) {
            xl[i] = p[(PK_BND + 2 * i) * 64];
            xu[i] = p[(PK_BND + 2 * i + 1) * 64];
        }
        xl[4] = xl[5] = p[(PK_BND + 8) * 64];
        xu[4] = xu[5] = 0.0;
    }
    template <int N> __device__ __forceinline__ void bounds(double (&)[N], double (&)[N]) const {}
    static constexpr bool on = true;
};
#endif
// Subspace-projected CG from the Cauchy point (TRON dspcg).  On entry s = Cauchy step;
// on exit x is the trial point and s = x - x0.
// w: on entry A s of the Cauchy step if w_valid (else it is computed here); on exit A s of the final step -- every
// return path leaves it current -- which dtron's predicted reduction needs next.
template <int N, class MA, class ML, class PK = NoPark>
EXA_BIG void spcg(double (&x)[N], double (&xl)[N], double (&xu)[N], const MA& A, ML& L,
                 double (&g)[N], double delta, double rtol, double (&s)[N], int itermax, double (&w)[N], bool w_valid,
                 unsigned long long* pc = nullptr, const PK& pk = PK())  // pc: diagnostic cycle counters (icfs, trpcg, prsrch, faces, cg its)
{
    double gf[N], id[N];
    if (!w_valid) symv<N>(A, s, w);
    pk.bounds(xl, xu);
    EXA_UNROLL
    for (int i = 0; i < N; i++) {
        double xi = x[i] + s[i];
        x[i] = fmax(xl[i], fmin(xi, xu[i]));
    }

    int iters = 0;
    for (int nfaces = 1; nfaces <= N; nfaces++) {
        unsigned fr = 0u;
        if (nfaces > 1) pk.bounds(xl, xu);
        EXA_UNROLL
        for (int i = 0; i < N; i++)
            if (xl[i] < x[i] && x[i] < xu[i]) fr |= (1u << i);
        if (fr == 0u) return;

        unsigned long long c0 = EXA_CLK(), c1;
        EXA_CNT(CNT_FACE);
        if (!PK::on) {
            icfs_masked<N>(fr, A, L, id);
            c1 = EXA_CLK();
            if (pc) pc[0] += c1 - c0;
        }

        double gfn2 = 0.0;
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            bool f = (fr >> i) & 1u;
            gf[i] = f ? (w[i] + g[i]) : 0.0;
            double gi = f ? g[i] : 0.0;
            gfn2 += gi * gi;
        }
        double gfnorm = sqrt(gfn2);
        if (PK::on) {
            // (same operations as above in another order: the factorisation does not read what the lines above produce)
            pk.put(PK_X, x); pk.put(PK_G, g); pk.put(PK_S, s);
            icfs_masked<N>(fr, A, L, id);
            c1 = EXA_CLK();
            if (pc) pc[0] += c1 - c0;
        }

        int itertr;
        c0 = EXA_CLK();
        int infotr = trpcg_masked<N>(fr, A, gf, delta, L, id, rtol * gfnorm, 0.0, itermax, w, itertr);
        iters += itertr;
        trsol_t<N>(L, id, w);
        c1 = EXA_CLK();
        if (pc) pc[1] += c1 - c0;
        if (PK::on) {
            pk.get(PK_X, x); pk.bounds(xl, xu); pk.get(PK_G, g); pk.get(PK_S, s);
        }

        prsrch_masked<N>(fr, x, xl, xu, A, gf, w);
        c0 = EXA_CLK();
        if (pc) {
            pc[2] += c0 - c1;
            pc[3] += 1;
            pc[4] += (unsigned long long)itertr;
        }

        EXA_UNROLL
        for (int i = 0; i < N; i++) s[i] += w[i];

        symv<N>(A, s, w);
        double gn2 = 0.0;
        EXA_UNROLL
        for (int i = 0; i < N; i++) {
            double gi = ((fr >> i) & 1u) ? (w[i] + g[i]) : 0.0;
            gn2 += gi * gi;
        }
#if defined(EXA_EXACT_ORACLE_ARITH)
        if (sqrt(gn2) <= rtol * gfnorm) return;
#else
        if (gn2 <= (rtol * rtol) * gfn2) return;   // ||g_free(new)|| <= rtol ||g_free||, compared as squares
#endif
        if (infotr == 3 || infotr == 4) return;
        if (iters > itermax) return;
    }
}

// Per-lane state of one trust-region Newton solve (isave/dsave of dtron + driver counters).
template <int N>
struct TronState {
    double f, fc, delta, alphac, prered, gts;   // gts = g's of the last model step (the ratio test needs it again)
    double xc[N], s[N];
    int iter, minor_iter, nfev;
#if defined(EXA_PROFILE_PHASES)
    unsigned long long pc[8];  // diagnostic: icfs, trpcg, prsrch cycles, faces, cg its, cauchy cycles, spcg cycles
#endif
};

// dtron, work = COMPUTE: model step from x (f, g, A valid at x).  x becomes the trial point.
// CLONES (device): the caller is a wavefront whose 64 lanes run the SAME solve; the Cauchy search then evaluates its candidate step
// lengths side by side (cauchy_clones; search_area: 2 N + 2 doubles of LDS).
template <int N, bool CLONES = false, class MA, class ML, class PK = NoPark>
EXA_HD void tron_compute(TronState<N>& st, double (&x)[N], double (&xl)[N], double (&xu)[N],
                         const MA& A, ML& L, double (&g)[N], double cgtol, int cg_itermax, double* search_area = nullptr, const PK& pk = PK())
{
#if defined(__HIP_DEVICE_COMPILE__)
#define EXA_CAUCHY(...) do { if constexpr (CLONES) cauchy_clones<N>(search_area, __VA_ARGS__); else cauchy<N>(__VA_ARGS__); } while (0)
#else
#define EXA_CAUCHY(...) cauchy<N>(__VA_ARGS__)
    (void)search_area;
#endif
    st.fc = st.f;
    EXA_UNROLL
    for (int i = 0; i < N; i++) st.xc[i] = x[i];
    if (PK::on) pk.put(PK_XC, st.xc);   // (only the ratio test needs the point again, to return to it)
    pk.bounds(xl, xu);
    double wa[N];   // A s: from the Cauchy search into dspcg, from dspcg into the predicted reduction (same products,
    bool wa_valid;  // same bits as recomputing them: two of a model step's ~ten matrix-vector products less)
#if defined(EXA_PROFILE_PHASES)
    unsigned long long c0 = EXA_CLK();
    EXA_CAUCHY(x, xl, xu, A, g, st.delta, st.alphac, st.s, wa, wa_valid);
    unsigned long long c1 = EXA_CLK();
    spcg<N>(x, xl, xu, A, L, g, st.delta, cgtol, st.s, cg_itermax, wa, wa_valid, st.pc, pk);
    st.pc[5] += c1 - c0;
    st.pc[6] += EXA_CLK() - c1;
#else
    EXA_CAUCHY(x, xl, xu, A, g, st.delta, st.alphac, st.s, wa, wa_valid);
    spcg<N>(x, xl, xu, A, L, g, st.delta, cgtol, st.s, cg_itermax, wa, wa_valid, nullptr, pk);
#endif
#undef EXA_CAUCHY
    st.gts = dot<N>(st.s, g);
    st.prered = -(st.gts + 0.5 * dot<N>(st.s, wa));
}

// dtron, work = EVALUATE: ratio test with ftrial = f(x).  Returns true when the step is
// accepted (caller must then refresh g, A at x); on rejection x is restored.
// task is set to TRON_RUN, TRON_CONV or TRON_WARN.
template <int N>
EXA_HD bool tron_evaluate(TronState<N>& st, double ftrial, double (&x)[N], const double (&g)[N], double frtol,
                          double fatol, double fmin_, int& task)
{
    const double eta0 = 1e-4, eta1 = 0.25, eta2 = 0.75;
    const double sigma1 = 0.25, sigma2 = 0.5, sigma3 = 4.0;
    double f = ftrial;
    const double fc = st.fc;
    double delta = st.delta;
    const double actred = fc - f;
    const double snorm = nrm2<N>(st.s);
    if (st.iter == 1) delta = fmin(delta, snorm);

    const double g0 = st.gts;   // = dot(g, s): the products commute, the sum runs in the same order
    const double curv = f - fc - g0;
    const double alpha = (curv <= 0.0) ? sigma3 : fmax(sigma1, -0.5 * exa_div(g0, curv));   // (division outside any branch: see trqsol)

    const double prered = st.prered;
    {
        // the four radius rules of dtron, all four formed and one selected (each is two or three operations; as nested
        // branches every rule costs a lone wavefront a VALU -> SALU -> VALU hand-over before it starts)
        const double as = alpha * snorm, s1d = sigma1 * delta, s2d = sigma2 * delta, s3d = sigma3 * delta;
        const double d0 = fmin(fmax(alpha, sigma1) * snorm, s2d);
        const double d1 = fmax(s1d, fmin(as, s2d));
        const double d2 = fmax(s1d, fmin(as, s3d));
        const double d3 = fmax(delta, fmin(as, s3d));
        const double d23 = (actred < eta2 * prered) ? d2 : d3;
        const double d123 = (actred < eta1 * prered) ? d1 : d23;
        delta = (actred < eta0 * prered) ? d0 : d123;
    }
    st.delta = delta;

    bool success = (actred > eta0 * prered);
    if (success) {
        st.iter++;
    } else {
        EXA_UNROLL
        for (int i = 0; i < N; i++) x[i] = st.xc[i];
        f = fc;
    }
    st.f = f;

    task = TRON_RUN;
    if (f < fmin_) task = TRON_WARN;
    if (fabs(actred) <= fatol && prered <= fatol) task = TRON_CONV;
    if (fabs(actred) <= frtol * fabs(f) && prered <= frtol * fabs(f)) task = TRON_CONV;
    return success;
}

}  // namespace exa
