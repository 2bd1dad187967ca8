// HIP kernels of the ADMM inner iteration for gfx950 (MI355X).
//
// Device layout (DESIGN.md "data layout in HBM"): every solution vector is ONE
// struct-of-arrays buffer of N = 2*Gp + 8*Lp doubles:
//     [ pg plane (Gp) | qg plane (Gp) | pij plane (Lp) | qij | pji | qji | wi | wj | ti | tj ]
// Gp, Lp = ngen, nline rounded up to 64.  Lane b of the branch kernel touches element b of
// each plane -> every load/store of a wavefront is one contiguous 512-byte segment, where the
// reference's interleaved 8-stride layout makes each wave touch 8x the cache lines.
// Padding elements hold u=v=z=l=lz=0, rho=1 and therefore contribute exact zeros everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include "branch_solver.h"
#include "genramp_solver.h"

namespace exa {

// loop control shared by the kernels of a fused inner loop (lives in device memory)
struct Ctrl {
    int done;         // set by the reduction (k_final_reduce / k_bus_group_t `red` / k_halo_unpack) when the inner loop must stop
    int inner;        // completed inner iterations
    int max_inner;
    int check_every;  // 1 = reference semantics
    double eps_pri;
    double beta;
    double res[4];    // primres, dualres, norm_z, mismatch of the last completed iteration
    double sums[4];   // their squares
    // speculative branch launch (exaadmm_run_inner / run_fixed on one GPU): iterations enqueued before the
    // current batch, and "the z/l pass of an iteration has run but its sums are not reduced yet"
    int base, pending;
    // z/l update fused into the branch launch (k_branch_t<MP, true>): major_iter of the launch that has NO pending
    // z/l update in front of it (the first iteration of the loop)
    int first_iter;
};

struct Dev {
    int G, L, B, Gp, Lp, N;
    // A partition (multi-GPU) stores the branches it owns first (from-bus local), then the replicas of the cut branches
    // a peer owns (to-bus local).  All L are solved and updated here; only the first Lown count in the residual norms.
    int Lown;
    double *u, *v, *z, *zp, *l, *lz, *rho;
    double *rp, *rd, *ax;
    const double* Y;    // 8 planes x Lp
    const double* bnd;  // 8 planes x Lp: vi lo, vi hi, vj lo, vj hi, ti lo, ti hi, tj lo, tj hi
    const double* rateA;
    double *ls1, *ls2, *mu;
    const double *pgmin, *pgmax, *qgmin, *qgmax, *c2, *c1;
    const double* c0;  // multi-period model only (the ramp subproblem's objective carries the constant)
    const double *Pd, *Qd, *YshR, *YshI;
    const double* busc;  // per bus, BUSC_N planes: what the bus update derives from rho and the shunt alone (k_bus_const)
    const int *FrStart, *FrIdx, *ToStart, *ToIdx, *GenStart, *GenIdx;
    const int *EndStart, *EndCode;  // per bus: generators, from-branches, to-branches (kind << 28 | index)
    // the same list laid out for the lane-group bus kernel: SlotCode[8 * bus + lane] = component of the bus's first chunk
    // served by that lane, EndCount[bus] = number of components -- one round of independent, coalesced loads instead of
    // the dependent pair EndStart -> EndCode (buses with more than 8 components use EndStart / EndCode for the rest)
    const int *SlotCode, *EndCount;
    // "push" layout of the fused loop (k_bus_rec): the branch / generator lanes drop, for every end on a local bus, the
    // 8 doubles the bus update needs from it -- (u+z, l, rho) of p and q and the two consensus terms of w and theta -- into
    // rec[8 * slot], slot = 8 * bus + position in the bus's component list (later chunks of a bus with more than 8
    // components: 8 * B + position in EndCode).  A bus lane's address then follows from its thread id alone: one
    // contiguous 64-byte read per component instead of an index load followed by 16 dependent 8-byte gathers.
    // rec_br[2 * branch + {0 from, 1 to}], rec_gen[generator] = slot or -1 (that end's bus belongs to a peer).
    double* rec;
    const int *rec_br, *rec_gen;
    // resident loop (k_resident): per branch (from-bus, its counts, to-bus, its counts), per generator (bus, counts);
    // counts = generators << 16 | components
    const int *br_bus4, *gen_bus2;
    int *st_auglag, *st_minor, *st_ticks;
    unsigned long long* st_cyc;  // diagnostic build (EXA_PROFILE_PHASES): 13 x Lp phase cycle counters
    unsigned long long* st_tl;   // diagnostic build: 128 timeline entries per wavefront (2 x Lp)
    double* partials;  // [blocks][4]
    // speculative branch launch: what the branch kernel READS (previous iterate; nullptr = the same buffers it writes)
    const double *u_in, *ls1_in, *ls2_in, *mu_in;
    // multi-GPU partition (dist_native.h): the ONE exchange of an iteration.  A bus-kernel lane that writes the v entries
    // of a cut branch's local end also drops them into the send buffer (halo_pos[branch] = position of the 4 doubles;
    // the component code carries END_CUT); the reduction block appends this rank's 4 partial sums to every peer's message.
    // halo_off: [nranks] position of the sums in the message TO peer p, then [nranks] in the message FROM peer p.
    int nranks, rank;
    const int* halo_pos;
    const int* halo_off;
    double* halo_send;
    const double* halo_recv;
    double* halo_local;   // this rank's own 4 partial sums
    // fused z/l variant: v the branch launch reads (the bus launch of the same parity writes d.v), per-block partial
    // sums of the branch launch's z/l prologue
    const double* v_in;
    double* partials_br;
    int npartials_br;
    // A partition with peers splits the branch launch of an iteration in two (exaadmm_hip.hip enqueue_dist_iterations): the
    // INTERIOR blocks (+ the generator blocks) start right after the bus launch, the blocks that hold cut branches wait for the
    // peers' halo.  blk0 = first branch block this launch covers, gen_part0 = partial-sum slot of the first generator block
    // (= number of branch blocks of the whole grid; a one-piece launch: blk0 = 0, gen_part0 = its own branch blocks).
    int blk0, gen_part0;
    // launches with more blocks than the chip has SIMDs (the multi-period model, grids beyond the resident loop): the tile of 64 branches that
    // branch block i solves -- heaviest tiles first (k_block_order).  Blocks start in index order; a tile whose straggler needs 9 trips and starts
    // in the launch's second round of wavefronts ends the launch a round late.  nullptr: block i solves tile i.
    const int* blk_order;
    // ... and runs TWO launches ahead of the stop test (the interior launch of iteration j+1 starts before the halo of iteration
    // j -- which carries the residual sums that decide about iteration j-1 -- has arrived): the iterate rotates through three
    // buffer sets, z and l included.  z_in / l_in = z, l of the iteration BEFORE the one whose z / l pass rides in this launch's
    // prologue (the pass reads them and writes d.z / d.l, another set); nullptr = in place (one GPU: two sets, d.zp kept).
    const double *z_in, *l_in;
    // multi-period model on the fused loop: what the ramp subproblem of a generator READS of the previous iterate (phat, s,
    // multiplier, penalty; nullptr = the buffers it writes), and how many per-period slots a block's partial residual sums
    // have (a block of 64 lanes may hold lanes of several periods: the norms are per period, max over periods)
    const double *r_u_in, *r_s_in, *r_mu_in, *r_xi_in;
    int part_k;
    // multi-period model (ModelMpacopf): T periods stored period-major inside every
    // plane (generator t*g1+i, branch t*l1+j, bus t*b1+k); T = 1 for the single-period model.
    // Ramp block (SolutionRamping of period t >= 1) at [t*g1 + i]; block 0 is unused.
    int T, g1, l1, b1;
    double *r_u, *r_s, *r_z, *r_zp, *r_l, *r_lz, *r_rho, *r_rp, *r_rd, *r_ax;
    double *r_mu, *r_xi;      // gen_membuf rows 7, 8: multiplier and penalty of the ramp constraint (persistent)
    const double* ramp_rate;  // per expanded generator: ramp_ratio * pgmax
};

__constant__ MuPowTable c_mupow;

// Hand-parked solver vectors (tron_device.h LanePark, round 6): behind the 40 rows of the lane code's LDS area that hold the parked constants / z, PK_SLOTS
// rows hold x, g, s around the factorisation / CG of every face, the ratio test's return point and the flows across the model step, and the box
// permanently -- 4 wavefronts per CU x 40 640 B of the CU's 160 KB.  What the compiler used to keep in AGPRs (k_resident 150 -> 58, its heavy form 170 -> 83,
// k_branch_t 140 -> 49; 599 -> 134 AGPR moves in k_branch_t) now costs LDS reads that issue beside the VALU: + 1.2 - 2.4 % at 13 659 buses (steady
// states and whole solve), + 1.6 % at 88 207 branches, - 1 - 2 % at 1 354 (profiles/r06_lds_parking.txt).  The host pass sees the no-op parker.
#if defined(__HIP_DEVICE_COMPILE__)
#define EXA_PARK_SLOTS PK_SLOTS
#define EXA_PARKER(p) LanePark{p}
#else
#define EXA_PARK_SLOTS PK_SLOTS
#define EXA_PARKER(p) NoPark()
#endif
constexpr int EW_BLOCK = 256;   // elementwise kernels
constexpr int BR_BLOCK = 64;    // branch kernel: one wavefront per block

__device__ __forceinline__ bool loop_done(const Ctrl* c) { return c != nullptr && c->done != 0; }

// ---------------------------------------------------------------- generator
// generator_kernel_two_level (acopf_generator_kernel_gpu.jl:1-22), one lane per generator
__device__ __forceinline__ void generator_update(const Dev& d, double baseMVA, int i, const double* vsrc, double& up, double& uq)
{
    const double* const v = vsrc ? vsrc : d.v;
    const int pg = i, qg = d.Gp + i;
    const double M = baseMVA;
    double t = (-(d.c1[i] * M + d.l[pg] + d.rho[pg] * (-v[pg] + d.z[pg]))) / (2 * d.c2[i] * (M * M) + d.rho[pg]);
    up = fmax(d.pgmin[i], fmin(d.pgmax[i], t));
    d.u[pg] = up;
    t = (-(d.l[qg] + d.rho[qg] * (-v[qg] + d.z[qg]))) / d.rho[qg];
    uq = fmax(d.qgmin[i], fmin(d.qgmax[i], t));
    d.u[qg] = uq;
}
__device__ __forceinline__ void generator_update(const Dev& d, double baseMVA, int i, const double* vsrc = nullptr)
{
    if (i >= d.G) return;
    double up, uq;
    generator_update(d, baseMVA, i, vsrc, up, uq);
}

// z_prev <- z ; z-update ; l-update ; residual terms of ONE element, the arithmetic of k_fused_zl
// (admm_inner_prestep + admm_update_z + admm_update_l + admm_update_residual) -- used by the branch launch's
// prologue when the z/l pass rides in it.  Returns the new l and z of the element.
__device__ __forceinline__ void zl_values(double u, double v, double rho, double beta, double lz, double zold, double lold,
                                          double (&s)[4], double& l_new, double& z_new)
{
    const double z = (-(lz + lold + rho * (u - v))) / (beta + rho);
    const double rp = u - v + z;
    const double rd = z - zold;
    const double ax = rp - z;
    l_new = -(lz + beta * z);
    z_new = z;
    s[0] += rp * rp;
    s[1] += rd * rd;
    s[2] += z * z;
    s[3] += ax * ax;
}
__device__ __forceinline__ void zl_element(const Dev& d, int e, double u, double v, double rho, double beta, double (&s)[4],
                                           double& l_new, double& z_new)
{
    const double lz = d.lz[e];
    const double zold = d.z[e];
    zl_values(u, v, rho, beta, lz, zold, d.l[e], s, l_new, z_new);
    d.zp[e] = zold;
    d.z[e] = z_new;
    d.l[e] = l_new;
}

// per-period partial sums of a wavefront whose lanes [first, first + 64) of a period-major index range (n per period) may belong to
// several periods: slot k of the block = period t_first + k.  t_lane < 0: the lane contributes nothing.
__device__ __forceinline__ void mp_block_partials(const Dev& d, int part, int first, int n, int total, int t_lane, const double (&zs)[4])
{
    const int t0 = first / n;
    const int last = min(total, first + BR_BLOCK) - 1;
    const int t1 = last >= first ? last / n : t0 - 1;
    for (int t = t0; t <= t1; t++) {
        double s[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s[j] = (t_lane == t) ? zs[j] : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) d.partials_br[((size_t)part * d.part_k + (t - t0)) * 4 + j] = s[j];
        }
    }
}

__global__ void k_generator(Dev d, double baseMVA, const Ctrl* ctrl)
{
    if (loop_done(ctrl)) return;
    generator_update(d, baseMVA, blockIdx.x * blockDim.x + threadIdx.x);
}

// Multi-period generator update (mpacopf_admm_update_x_cpu.jl:5-31): period 0 keeps the closed
// form; a generator of period t >= 1 solves the 3-variable ramp subproblem (p_t, phat_{t-1}, s_t)
// by augmented Lagrangian x TRON (auglag_generator_kernel, mpacopf_auglag_generator_kernel_cpu.jl),
// one lane per generator and period.
__device__ __forceinline__ void gen_ramp_lane(const Dev& d, const BranchOptions& o, double baseMVA, int major_iter, int i)
{
    if (i >= d.G) return;
    if (i < d.g1) {
        generator_update(d, baseMVA, i);
        return;
    }
    const int pg = i, qg = d.Gp + i;
    d.u[qg] = fmax(d.qgmin[i], fmin(d.qgmax[i], (-(d.l[qg] + d.rho[qg] * (-d.v[qg] + d.z[qg]))) / d.rho[qg]));
    double x[3], xl[3], xu[3];
    xl[0] = xl[1] = d.pgmin[i];
    xu[0] = xu[1] = d.pgmax[i];
    xl[2] = -d.ramp_rate[i];
    xu[2] = d.ramp_rate[i];
    x[0] = fmin(xu[0], fmax(xl[0], d.u[pg]));
    x[1] = fmin(xu[1], fmax(xl[1], d.r_u[i]));
    x[2] = fmin(xu[2], fmax(xl[2], d.r_s[i]));
    GenRampData gd;
    gd.lam_p = d.l[pg];
    gd.lam_h = d.r_l[i];
    gd.rho_p = d.rho[pg];
    gd.rho_h = d.r_rho[i];
    gd.xt_p = d.v[pg] - d.z[pg];
    gd.xt_h = d.v[pg - d.g1] - d.r_z[i];   // pbar of the previous period (mpacopf_admm_update_x_cpu.jl:19)
    gd.mu = d.r_mu[i];
    gd.xi = (major_iter <= 1) ? 10.0 : d.r_xi[i];
    gd.c2 = d.c2[i];
    gd.c1 = d.c1[i];
    gd.c0 = d.c0[i];
    gd.M = baseMVA;
    int it, minor;
    genramp_solve(gd, o, c_mupow, x, xl, xu, it, minor);
    d.u[pg] = x[0];
    d.r_u[i] = x[1];
    d.r_s[i] = x[2];
    d.r_mu[i] = gd.mu;
    d.r_xi[i] = gd.xi;
}

// The same lane in the fused loop (k_branch_t<true, true>): the z / l / residual pass of the PREVIOUS iteration for the lane's own
// elements -- pg, qg and, for a generator of period t >= 1, its ramp element (consensus partner: pbar of the previous period,
// mpacopf_admm_update_z_cpu.jl:10-14) -- then the update, reading the previous iterate (u_in, v_in, r_*_in) and writing the
// other buffer set.  zs receives the lane's residual terms (they count in the lane's period).
__device__ __forceinline__ void gen_ramp_lane_zl(const Dev& d, const BranchOptions& o, double baseMVA, int major_iter, int i, bool do_zl,
                                                 double zl_beta, double (&zs)[4])
{
    if (i >= d.G) return;
    const double* const uin = d.u_in ? d.u_in : d.u;
    const double* const vin = d.v_in ? d.v_in : d.v;
    const int pg = i, qg = d.Gp + i;
    const bool ramp = (i >= d.g1);
    if (do_zl) {
        double ln, zn;
        zl_element(d, pg, uin[pg], vin[pg], d.rho[pg], zl_beta, zs, ln, zn);
        zl_element(d, qg, uin[qg], vin[qg], d.rho[qg], zl_beta, zs, ln, zn);
        if (ramp) {
            const double lz = d.r_lz[i], zold = d.r_z[i];
            zl_values((d.r_u_in ? d.r_u_in : d.r_u)[i], vin[pg - d.g1], d.r_rho[i], zl_beta, lz, zold, d.r_l[i], zs, ln, zn);
            d.r_zp[i] = zold;
            d.r_z[i] = zn;
            d.r_l[i] = ln;
        }
    }
    if (!ramp) {
        double up, uq;
        generator_update(d, baseMVA, i, vin, up, uq);
        if (d.rec != nullptr) {   // what the bus update needs from this generator (the fields of k_branch_t's single-period generator lanes)
            double* q = d.rec + 8 * (size_t)d.rec_gen[i];
            q[0] = up + d.z[pg];
            q[1] = d.l[pg];
            q[2] = d.rho[pg];
            q[3] = uq + d.z[qg];
            q[4] = d.l[qg];
            q[5] = d.rho[qg];
        }
        return;
    }
    d.u[qg] = fmax(d.qgmin[i], fmin(d.qgmax[i], (-(d.l[qg] + d.rho[qg] * (-vin[qg] + d.z[qg]))) / d.rho[qg]));
    double x[3], xl[3], xu[3];
    xl[0] = xl[1] = d.pgmin[i];
    xu[0] = xu[1] = d.pgmax[i];
    xl[2] = -d.ramp_rate[i];
    xu[2] = d.ramp_rate[i];
    x[0] = fmin(xu[0], fmax(xl[0], uin[pg]));
    x[1] = fmin(xu[1], fmax(xl[1], (d.r_u_in ? d.r_u_in : d.r_u)[i]));
    x[2] = fmin(xu[2], fmax(xl[2], (d.r_s_in ? d.r_s_in : d.r_s)[i]));
    GenRampData gd;
    gd.lam_p = d.l[pg];
    gd.lam_h = d.r_l[i];
    gd.rho_p = d.rho[pg];
    gd.rho_h = d.r_rho[i];
    gd.xt_p = vin[pg] - d.z[pg];
    gd.xt_h = vin[pg - d.g1] - d.r_z[i];   // pbar of the previous period (mpacopf_admm_update_x_cpu.jl:19)
    gd.mu = (d.r_mu_in ? d.r_mu_in : d.r_mu)[i];
    gd.xi = (major_iter <= 1) ? 10.0 : (d.r_xi_in ? d.r_xi_in : d.r_xi)[i];
    gd.c2 = d.c2[i];
    gd.c1 = d.c1[i];
    gd.c0 = d.c0[i];
    gd.M = baseMVA;
    int it, minor;
    genramp_solve(gd, o, c_mupow, x, xl, xu, it, minor);
    d.u[pg] = x[0];
    d.r_u[i] = x[1];
    d.r_s[i] = x[2];
    d.r_mu[i] = gd.mu;
    d.r_xi[i] = gd.xi;
    if (d.rec != nullptr) {
        double* q = d.rec + 8 * (size_t)d.rec_gen[i];
        q[0] = x[0] + d.z[pg];
        q[1] = d.l[pg];
        q[2] = d.rho[pg];
        q[3] = d.u[qg] + d.z[qg];
        q[4] = d.l[qg];
        q[5] = d.rho[qg];
        // ... and what the bus of the SAME generator one period earlier needs from this lane's ramp element (its real power is tied
        // to it, bus_kernel_ramp, mpacopf_bus_kernel_cpu.jl:53-62): into the two free fields of that generator's record
        double* qp = d.rec + 8 * (size_t)d.rec_gen[i - d.g1];
        qp[6] = d.r_l[i] + d.r_rho[i] * (x[1] + d.r_z[i]);
        qp[7] = d.r_rho[i];
    }
}

__global__ void __launch_bounds__(64) k_gen_ramp(Dev d, BranchOptions o, double baseMVA, int major_iter, const Ctrl* ctrl)
{
    if (ctrl != nullptr) {
        if (ctrl->done) return;
        major_iter = ctrl->inner + 1;
    }
    gen_ramp_lane(d, o, baseMVA, major_iter, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------- branch
// auglag_linelimit_two_level_alternative (acopf_auglag_linelimit_kernel_gpu.jl:1-151):
// one lane per branch, 64 branches per wavefront, one wavefront per SIMD.  The packed 6x6 Hessian, its Cholesky factor and
// the vectors / scalars the solver is working on live in registers (256 VGPR + 49 AGPRs of spill space); the wavefront's 40 KB
// of LDS hold, [row][lane], what a phase does not touch: the 32 read-only problem constants (l, rho, v-z, Y) across the model
// step (round 3), and -- round 6, tron_device.h LanePark -- x, g, s across the factorisation / CG, the ratio test's return point and
// the flows across the model step, the box permanently.  (The reference keeps 1728 B of shared memory per BRANCH and one 32-thread
// block per branch.  Variants measured and rejected -- matrices staged in LDS, constants read from LDS at every use, two waves per
// SIMD in three forms, fewer branches per wavefront, lane regrouping, persistent wavefronts: profiles/archive_r01_r02/
// r02_two_waves_per_simd_experiments.txt, r02_lanes_per_wave.txt, r02_wave_order_experiment.txt, profiles/r06_two_waves_per_simd.txt,
// r06_persistent_tiles.txt.)
//
// The Dev block is read through a pointer (not by-value kernel arguments): its ~40 pointers would
// otherwise sit in SGPRs across the whole solver and spill.
// MP = true: the generator blocks solve the ramp subproblems of the multi-period model (separate
// instantiation so that the single-period kernel's code and register allocation stay untouched).
// ZL = true (the speculative loop, single-period model): launch of iteration ctrl->base + spec_j + 1.
// The z / l / residual pass of the PREVIOUS iteration rides in this launch's prologue -- every lane already
// loads u, v, z, l, rho of its own 8 (generator lanes: 2) elements, so the pass costs 8 more loads (lz) and
// its arithmetic instead of a kernel of its own (6 reads + 3 writes per element and a launch); per-block
// partial sums of the four residual norms go to partials_br, the bus launch that follows reduces them
// (k_bus_group_t, `red`).  The launch is speculative: it does not wait for the loop test of the previous
// iteration; it reads the previous iterate (u_in, v_in, ls*_in) and writes the other buffer set, so a stop
// detected one launch later leaves the iterate of the stopping iteration -- including the z, l this prologue
// has just completed for it -- intact.
// A partition (multi-GPU) solves the cut branches it shares with a peer as well (bit-identical replicas on
// both owners): branches at or past d.Lown run the same prologue and solve, but add nothing to the residual sums.
// (EXA_BR_WAVES: experiment switch, profiles/r05_two_waves_budget.txt, r06_two_waves_per_simd.txt -- wavefronts per SIMD the register allocator must leave room for)
#ifndef EXA_BR_WAVES
#define EXA_BR_WAVES 1
#endif
template <bool MP, bool ZL = false>
__global__ void __launch_bounds__(BR_BLOCK, EXA_BR_WAVES) k_branch_t(const Dev* __restrict__ dp, BranchOptions o, double scale, int major_iter, Ctrl* ctrl,
                                                           int gen_blocks, int spec_j, double baseMVA)   // (all inside the 16 preloaded SGPRs)
{
#if defined(EXA_PROFILE_PHASES)
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();   // first instruction of the wave
#endif
    const Dev& d = *dp;
    bool do_zl = false;
    double zl_beta = 0.0;
    if (ZL) {
        if (ctrl->done) return;
        major_iter = ctrl->base + spec_j + 1;
        do_zl = (major_iter != ctrl->first_iter);
        zl_beta = ctrl->beta;
        if (do_zl && blockIdx.x == 0 && threadIdx.x == 0) ctrl->pending = 1;   // read by the NEXT launches only (bus `red`, k_halo_unpack)
    } else if (ctrl != nullptr) {
        if (ctrl->done) return;
        major_iter = ctrl->inner + 1;
    }
    // blocks past the branch range do the generator update (independent of the branches; one
    // launch and one inter-kernel gap less per iteration).  gen_blocks = 0: branches only.
    const int nbr_blocks = gridDim.x - gen_blocks;
    // (with Dev::blk_order -- more blocks than SIMDs -- the generator / ramp blocks are the launch's FIRST blocks: behind 1 280 branch blocks they
    //  started in its second round of wavefronts.  bid: the block's number in the layout "branch blocks, then generator blocks")
    const bool gen_first = ZL && d.blk_order != nullptr;
    const int bid = gen_first ? (((int)blockIdx.x < gen_blocks) ? nbr_blocks + (int)blockIdx.x : (int)blockIdx.x - gen_blocks) : (int)blockIdx.x;
    double zs[4] = {0.0, 0.0, 0.0, 0.0};   // ZL: this lane's residual terms
    if (bid >= nbr_blocks) {
        const int gi = (bid - nbr_blocks) * BR_BLOCK + threadIdx.x;
        if (MP && ZL) {   // multi-period model on the fused loop: z / l pass (incl. the ramp element) + ramp subproblem, per-period partial sums
            gen_ramp_lane_zl(d, o, baseMVA, major_iter, gi, do_zl, zl_beta, zs);
            if (do_zl) mp_block_partials(d, d.gen_part0 + (bid - nbr_blocks), (bid - nbr_blocks) * BR_BLOCK, d.g1, d.G, gi < d.G ? gi / d.g1 : -1, zs);
        } else if (MP)   // multi-period model: ramp subproblems run beside the branch solves, same launch
            gen_ramp_lane(d, o, baseMVA, major_iter, gi);
        else if (ZL) {
            const double* const uin = d.u_in ? d.u_in : d.u;
            const double* const vin = d.v_in ? d.v_in : d.v;
            const double* const zsrc = (do_zl && d.z_in) ? d.z_in : d.z;
            const double* const lsrc = (do_zl && d.l_in) ? d.l_in : d.l;
            if (gi < d.G) {
                const int pg = gi, qg = d.Gp + gi;
                const double rho_p = d.rho[pg], rho_q = d.rho[qg];
                double lp = lsrc[pg], lq = lsrc[qg], zp = zsrc[pg], zq = zsrc[qg];
                if (do_zl) {
                    const double lzp = d.lz[pg], lzq = d.lz[qg];
                    const double zold_p = zp, zold_q = zq;
                    zl_values(uin[pg], vin[pg], rho_p, zl_beta, lzp, zold_p, lp, zs, lp, zp);
                    zl_values(uin[qg], vin[qg], rho_q, zl_beta, lzq, zold_q, lq, zs, lq, zq);
                    d.zp[pg] = zold_p;
                    d.z[pg] = zp;
                    d.l[pg] = lp;
                    d.zp[qg] = zold_q;
                    d.z[qg] = zq;
                    d.l[qg] = lq;
                }
                double up, uq;
                generator_update(d, baseMVA, gi, vin, up, uq);
                if (d.rec != nullptr) {   // what the bus update needs from this generator, where its bus lane will look for it
                    double* q = d.rec + 8 * (size_t)d.rec_gen[gi];
                    q[0] = up + zp;
                    q[1] = lp;
                    q[2] = rho_p;
                    q[3] = uq + zq;
                    q[4] = lq;
                    q[5] = rho_q;
                }
            }
            if (do_zl) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) zs[j] += __shfl_down(zs[j], off, 64);
                }
                if (threadIdx.x == 0) {
                    const int part = d.gen_part0 + (bid - nbr_blocks);
#pragma unroll
                    for (int j = 0; j < 4; j++) d.partials_br[part * 4 + j] = zs[j];
                }
            }
        } else
            generator_update(d, baseMVA, gi);
        return;
    }
    // mu^0.1, mu^0.9 table: the lanes index it (branch_solver.h mu_pow).  Filled before any lane leaves (the last block
    // of the step-wise launch has fewer than 64 lanes).
    __shared__ MuPowTable pw_lds;
    if (threadIdx.x < 24) reinterpret_cast<double*>(&pw_lds)[threadIdx.x] = reinterpret_cast<const double*>(&c_mupow)[threadIdx.x];
    const int blk = (ZL ? d.blk0 : 0) + ((ZL && d.blk_order != nullptr) ? d.blk_order[bid] : bid);   // (split launches of a partition start at block blk0)
    const int slot = blk * BR_BLOCK + threadIdx.x;
    const bool lane_on = slot < d.L;
    if (!ZL && !lane_on) return;
#if defined(EXA_PROFILE_PHASES)
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int b = lane_on ? slot : 0;   // (ZL: lanes past the range read branch 0, write nothing)
    const int Lp = d.Lp;
    const int base = 2 * d.Gp + b;
    const double* const uin = d.u_in ? d.u_in : d.u;
    const double* const vin = (ZL && d.v_in) ? d.v_in : d.v;

    // LDS, [entry][lane]: the branch's 32 read-only constants, parked across the model steps (branch_solver.h), and -- fused
    // loop -- the 8 z entries of the iterate for the epilogue
    __shared__ double park_lds[(40 + EXA_PARK_SLOTS) * BR_BLOCK];
    double ub[8], bnd[8];
    BranchData bd;
    double rate_0;
    int rec_from = -1, rec_to = -1;
    if (ZL) {
        // Every load of the prologue first, every store after them.  As far as the compiler knows the planes may alias,
        // so a store between two loads is a full memory round trip: with the z/l pass written element by element
        // (load, compute, store, eight times, then the multipliers) a wave needed 10.6 k clocks from kernel entry to
        // "loads landed" against 4.0 k for the launch without the pass (profiles/archive_r01_r02/r02_wave_timeline_*).
        double vv[8], zo[8], lo[8], lzv[8];
        const double* const zsrc = (do_zl && d.z_in) ? d.z_in : d.z;   // (three-set rotation of a partition with peers: see Dev::z_in)
        const double* const lsrc = (do_zl && d.l_in) ? d.l_in : d.l;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int e = base + k * Lp;
            ub[k] = uin[e];
            bd.rho_[k] = d.rho[e];
            vv[k] = vin[e];
            zo[k] = zsrc[e];
            lo[k] = lsrc[e];
            lzv[k] = do_zl ? d.lz[e] : 0.0;
            bd.Y_[k] = d.Y[b + k * Lp];
            bnd[k] = d.bnd[b + k * Lp];
        }
        bd.ls1 = (d.ls1_in ? d.ls1_in : d.ls1)[b];
        bd.ls2 = (d.ls2_in ? d.ls2_in : d.ls2)[b];
        const double mu_0 = (d.mu_in ? d.mu_in : d.mu)[b];
        rate_0 = d.rateA[b];
        if (d.rec != nullptr) {   // where this branch's two bus records go: asked for now, not behind the solve (a memory round
            rec_from = d.rec_br[2 * b];       // trip in front of the stores of the launch's slowest wavefront)
            rec_to = d.rec_br[2 * b + 1];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (do_zl && lane_on) {
                const int e = base + k * Lp;
                double zn;
                zl_values(ub[k], vv[k], bd.rho_[k], zl_beta, lzv[k], zo[k], lo[k], zs, bd.lam_[k], zn);
                d.zp[e] = zo[k];
                d.z[e] = zn;
                d.l[e] = bd.lam_[k];
                bd.xt_[k] = vv[k] - zn;
                zo[k] = zn;
            } else {
                bd.lam_[k] = lo[k];
                bd.xt_[k] = vv[k] - zo[k];
            }
            park_lds[(32 + k) * BR_BLOCK + threadIdx.x] = zo[k];   // z of the iterate: the epilogue's bus records need it again
        }
        if (MP && do_zl) {   // multi-period model: per-period partial sums (a block may hold lanes of several periods)
            mp_block_partials(d, blk, blk * BR_BLOCK, d.l1, d.L, lane_on ? b / d.l1 : -1, zs);
        } else if (do_zl) {   // per-block partial sums of the four residual norms (fixed topology: wave shuffle, one wave per block)
            const bool counts = (b < d.Lown);   // a peer's branch replicated here: its terms are summed by its owner
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!counts) zs[j] = 0.0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) zs[j] += __shfl_down(zs[j], off, 64);
            }
            if (threadIdx.x == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) d.partials_br[blk * 4 + j] = zs[j];
            }
        }
        if (!lane_on) return;
        bd.mu = (major_iter == 1) ? 10.0 : mu_0;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int e = base + k * Lp;
            ub[k] = uin[e];
            bd.rho_[k] = d.rho[e];
            bd.lam_[k] = d.l[e];
            bd.xt_[k] = vin[e] - d.z[e];
            bd.Y_[k] = d.Y[b + k * Lp];
            bnd[k] = d.bnd[b + k * Lp];
        }
        bd.ls1 = (d.ls1_in ? d.ls1_in : d.ls1)[b];
        bd.ls2 = (d.ls2_in ? d.ls2_in : d.ls2)[b];
        bd.mu = (major_iter == 1) ? 10.0 : (d.mu_in ? d.mu_in : d.mu)[b];
        rate_0 = d.rateA[b];
    }
    bd.scale = scale;
    SymMat<6> A, L;

    double x[6], xl[6], xu[6], phi[4];
    branch_start(ub, bnd, rate_0, x, xl, xu);
    BranchStats st;
#if defined(EXA_PROFILE_PHASES)
    st.tl = d.st_tl ? d.st_tl + (size_t)bid * 128 : nullptr;
    // force the loads to land before the stamp: consume one loaded value in the asm dependency
    unsigned long long t_loaded;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_loaded) : "v"(x[0]) : "memory");
#endif
    branch_solve(bd, o, pw_lds, x, xl, xu, A, L, phi, st, park_lds + threadIdx.x, nullptr, EXA_PARKER(park_lds + 40 * BR_BLOCK + threadIdx.x));
#if defined(EXA_PROFILE_PHASES)
    const unsigned long long t_solved = __builtin_amdgcn_s_memtime();
#endif

    d.u[base] = phi[0];
    d.u[base + Lp] = phi[1];
    d.u[base + 2 * Lp] = phi[2];
    d.u[base + 3 * Lp] = phi[3];
    d.u[base + 4 * Lp] = x[0] * x[0];
    d.u[base + 5 * Lp] = x[1] * x[1];
    d.u[base + 6 * Lp] = x[2];
    d.u[base + 7 * Lp] = x[3];
    if (ZL && d.rec != nullptr) {
        // "push": what the bus update needs from the two ends of this branch -- (u+z, l, rho) of p and q, and the terms
        // l + rho (u + z) of w and theta, evaluated exactly as the bus kernels evaluate them -- goes where the lanes of the
        // end's bus will read it with an address that follows from their thread id (k_bus_rec)
        double zz[8];
#pragma unroll
        for (int k = 0; k < 8; k++) zz[k] = park_lds[(32 + k) * BR_BLOCK + threadIdx.x];
        const double un[8] = {phi[0], phi[1], phi[2], phi[3], x[0] * x[0], x[1] * x[1], x[2], x[3]};
        const int rf = rec_from, rt = rec_to;
        if (rf >= 0) {
            double* q = d.rec + 8 * (size_t)rf;
            q[0] = un[0] + zz[0];  q[1] = bd.lam_[0];  q[2] = bd.rho_[0];
            q[3] = un[1] + zz[1];  q[4] = bd.lam_[1];  q[5] = bd.rho_[1];
            q[6] = bd.lam_[4] + bd.rho_[4] * (un[4] + zz[4]);
            q[7] = bd.lam_[6] + bd.rho_[6] * (un[6] + zz[6]);
        }
        if (rt >= 0) {
            double* q = d.rec + 8 * (size_t)rt;
            q[0] = un[2] + zz[2];  q[1] = bd.lam_[2];  q[2] = bd.rho_[2];
            q[3] = un[3] + zz[3];  q[4] = bd.lam_[3];  q[5] = bd.rho_[3];
            q[6] = bd.lam_[5] + bd.rho_[5] * (un[5] + zz[5]);
            q[7] = bd.lam_[7] + bd.rho_[7] * (un[7] + zz[7]);
        }
    }
    d.ls1[b] = bd.ls1;
    d.ls2[b] = bd.ls2;
    d.mu[b] = bd.mu;
    d.st_auglag[b] = st.auglag_it;
    d.st_minor[b] = st.minor_it;
    d.st_ticks[b] = st.ticks;
#if defined(EXA_PROFILE_PHASES)
    for (int q = 0; q < 13; q++) d.st_cyc[q * d.Lp + b] = st.cyc[q];
    if (st.tl) {   // per-wave prologue / epilogue stamps (entries 123..127 of the timeline)
        unsigned long long t_stored;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_stored) : : "memory");
        {   // where the wavefront ran: HW_ID (wave / SIMD / CU / SH / SE) and the XCC -- scripts/diag_wave_starts.py pairs up successors on a SIMD
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
            st.tl[121] = ((unsigned long long)xcc << 32) | hw;
        }
        st.tl[122] = t_start;
        st.tl[123] = t_entry;
        st.tl[124] = t_loaded;
        st.tl[125] = t_solved;
        st.tl[126] = t_stored;
    }
#endif
}

// Multi-period model: the real power of a generator of period t < T-1 is also tied to the copy
// phat kept by period t+1 (ramp block index pg + g1); the last period has no such partner.
__device__ __forceinline__ bool bus_gen_has_ramp(const Dev& d, int pg) { return d.T > 1 && pg < (d.T - 1) * d.g1; }

// ---------------------------------------------------------------- bus
// bus_kernel_two_level_alternative (acopf_bus_kernel_gpu.jl:1-119), one lane per bus.
// Gathers 4 planes per incident branch end, 2 per generator; scatters v to the same places.
__global__ void k_bus(Dev d, double baseMVA, const Ctrl* ctrl)
{
    if (loop_done(ctrl)) return;
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= d.B) return;
    const int Lp = d.Lp, Gp = d.Gp, lb = 2 * d.Gp;
    const double *u = d.u, *z = d.z, *l = d.l, *rho = d.rho;
    double* v = d.v;

    double common_wi = 0.0, common_ti = 0.0, inv_rhosum_p = 0.0, inv_rhosum_q = 0.0;
    double rhosum_w = 0.0, rhosum_t = 0.0;
    const int f0 = d.FrStart[I], f1 = d.FrStart[I + 1];
    const int t0 = d.ToStart[I], t1 = d.ToStart[I + 1];
    const int g0 = d.GenStart[I], g1 = d.GenStart[I + 1];

    for (int k = f0; k < f1; k++) {
        const int e = lb + d.FrIdx[k];
        const int ew = e + 4 * Lp, et = e + 6 * Lp;
        common_wi += l[ew] + rho[ew] * (u[ew] + z[ew]);
        common_ti += l[et] + rho[et] * (u[et] + z[et]);
        inv_rhosum_p += 1.0 / rho[e];
        inv_rhosum_q += 1.0 / rho[e + Lp];
        rhosum_w += rho[ew];
        rhosum_t += rho[et];
    }
    for (int k = t0; k < t1; k++) {
        const int e = lb + d.ToIdx[k];
        const int ew = e + 5 * Lp, et = e + 7 * Lp;
        common_wi += l[ew] + rho[ew] * (u[ew] + z[ew]);
        common_ti += l[et] + rho[et] * (u[et] + z[et]);
        inv_rhosum_p += 1.0 / rho[e + 2 * Lp];
        inv_rhosum_q += 1.0 / rho[e + 3 * Lp];
        rhosum_w += rho[ew];
        rhosum_t += rho[et];
    }
    common_wi /= rhosum_w;

    double rhs1 = 0.0, rhs2 = 0.0, inv_rhosum_pg = 0.0, inv_rhosum_qg = 0.0;
    for (int k = g0; k < g1; k++) {
        const int pg = d.GenIdx[k], qg = Gp + pg;
        if (bus_gen_has_ramp(d, pg)) {  // bus_kernel_ramp (mpacopf_bus_kernel_cpu.jl:53-62)
            const int ri = pg + d.g1;
            const double den = rho[pg] + d.r_rho[ri];
            rhs1 += ((l[pg] + rho[pg] * (u[pg] + z[pg])) + (d.r_l[ri] + d.r_rho[ri] * (d.r_u[ri] + d.r_z[ri]))) / den;
            inv_rhosum_pg += 1.0 / den;
        } else {
            rhs1 += (u[pg] + z[pg]) + (l[pg] / rho[pg]);
            inv_rhosum_pg += 1.0 / rho[pg];
        }
        rhs2 += (u[qg] + z[qg]) + (l[qg] / rho[qg]);
        inv_rhosum_qg += 1.0 / rho[qg];
    }
    rhs1 -= (d.Pd[I] / baseMVA);
    rhs2 -= (d.Qd[I] / baseMVA);
    for (int k = f0; k < f1; k++) {
        const int e = lb + d.FrIdx[k], eq = e + Lp;
        rhs1 -= (u[e] + z[e]) + (l[e] / rho[e]);
        rhs2 -= (u[eq] + z[eq]) + (l[eq] / rho[eq]);
    }
    for (int k = t0; k < t1; k++) {
        const int e = lb + d.ToIdx[k] + 2 * Lp, eq = e + Lp;
        rhs1 -= (u[e] + z[e]) + (l[e] / rho[e]);
        rhs2 -= (u[eq] + z[eq]) + (l[eq] / rho[eq]);
    }
    const double ysr = d.YshR[I], ysi = d.YshI[I];
    rhs1 -= ysr * common_wi;
    rhs2 += ysi * common_wi;

    const double A11 = (inv_rhosum_pg + inv_rhosum_p) + (ysr * ysr / rhosum_w);
    const double A12 = -ysr * (ysi / rhosum_w);
    const double A21 = A12;
    const double A22 = (inv_rhosum_qg + inv_rhosum_q) + (ysi * ysi / rhosum_w);
    const double mu2 = (rhs2 - (A21 / A11) * rhs1) / (A22 - (A21 / A11) * A12);
    const double mu1 = (rhs1 - A12 * mu2) / A11;
    const double wi = common_wi + ((ysr * mu1 - ysi * mu2) / rhosum_w);
    const double ti = common_ti / rhosum_t;

    for (int k = g0; k < g1; k++) {
        const int pg = d.GenIdx[k], qg = Gp + pg;
        if (bus_gen_has_ramp(d, pg)) {  // mpacopf_bus_kernel_cpu.jl:98-101
            const int ri = pg + d.g1;
            v[pg] = ((l[pg] + rho[pg] * (u[pg] + z[pg])) + (d.r_l[ri] + d.r_rho[ri] * (d.r_u[ri] + d.r_z[ri])) - mu1) /
                    (rho[pg] + d.r_rho[ri]);
        } else {
            v[pg] = (u[pg] + z[pg]) + (l[pg] - mu1) / rho[pg];
        }
        v[qg] = (u[qg] + z[qg]) + (l[qg] - mu2) / rho[qg];
    }
    for (int k = f0; k < f1; k++) {
        const int e = lb + d.FrIdx[k], eq = e + Lp;
        v[e] = (u[e] + z[e]) + (l[e] + mu1) / rho[e];
        v[eq] = (u[eq] + z[eq]) + (l[eq] + mu2) / rho[eq];
        v[e + 4 * Lp] = wi;
        v[e + 6 * Lp] = ti;
    }
    for (int k = t0; k < t1; k++) {
        const int e0 = lb + d.ToIdx[k];
        const int e = e0 + 2 * Lp, eq = e0 + 3 * Lp;
        v[e] = (u[e] + z[e]) + (l[e] + mu1) / rho[e];
        v[eq] = (u[eq] + z[eq]) + (l[eq] + mu2) / rho[eq];
        v[e0 + 5 * Lp] = wi;
        v[e0 + 7 * Lp] = ti;
    }
}

// ---------------------------------------------------------------- bus, lane-group version
// Same arithmetic, in the same order, as k_bus / bus_kernel_two_level_alternative
// (acopf_bus_kernel_gpu.jl:1-119), organised for latency: BUS_W lanes serve one bus.
//   * every lane fetches ONE incident component (generator / from-end / to-end) of the bus --
//     the gathers of a bus are issued in parallel instead of one dependent loop per lane;
//   * each lane evaluates the TERMS the reference adds up, exactly as written there, and parks them in
//     LDS ([term][lane] inside the group); the group leader adds them in the reference's order (generators,
//     -Pd, from-ends, to-ends), so the floating-point result is identical to the one-lane-per-bus kernel;
//   * everything that depends on rho and the shunt only -- the sums of rho and of 1/rho over the bus's
//     components, the 2x2 matrix A11, A12, A22 built from them, A21/A11 and the pivot A22 - (A21/A11) A12 -- is
//     constant while rho is (it is, inside a solve): k_bus_const evaluates it once per rho upload, in the
//     reference's order, and the leader reads six doubles per bus instead of re-deriving them from 2 rho gathers per
//     component, five divisions shorter;
//   * the leader publishes mu1, mu2, w, theta through LDS and every lane writes the v entries
//     of its own component from the registers it already holds.
// Buses with more than BUS_W incident components are processed in chunks.
constexpr int BUS_W = 8;         // measured at 13 659 buses (kernel-trace average): 8 -> 12.8, 16 -> 13.5, 32 -> 15.5 us
constexpr int BUS_BLOCK = 256;   // measured at 13 659 buses: 64 -> 15.1, 128 -> 14.5, 256 -> 12.4, 512 -> 11.9, 1024 -> 13.2 us; 512 loses 5 % of the non-branch time at 1M branches
static_assert(BUS_W == 8, "k_bus_group_t reads its group's byte of a wavefront ballot and unrolls the leader's sums over 8 slots");
static_assert(BUS_BLOCK == EW_BLOCK, "the reduction block of k_bus_group_t uses block_reduce4, which is laid out for EW_BLOCK threads");
constexpr int BUS_REC = 16;  // doubles per record: (u,z,l,rho) x (p,q,w,theta) or (u,z,l,rho) x (pg,qg)
enum { END_GEN = 0, END_FROM = 1, END_TO = 2 };
constexpr int END_CUT = 1 << 30;        // component code flag: the branch is shared with a peer partition (its v entries go into the halo message)
constexpr int END_INDEX = 0x0fffffff;
// planes of Dev::busc: the rho / shunt constants, then the per-unit load and the shunt themselves (so that everything a bus
// lane needs beside the records comes from ONE array whose address is among the preloaded kernel arguments)
constexpr int BUSC_A11 = 0, BUSC_A12 = 1, BUSC_R21 = 2, BUSC_PIV = 3, BUSC_RW = 4, BUSC_RT = 5, BUSC_PD = 6, BUSC_QD = 7, BUSC_YSR = 8,
              BUSC_YSI = 9, BUSC_N = 10;

__device__ __forceinline__ void bus_load_end(const Dev& d, int code, double (&r)[BUS_REC], int (&idx)[4])
{
    const int kind = (code >> 28) & 3, c = code & END_INDEX;
    const int Lp = d.Lp;
    if (kind == END_GEN) {
        idx[0] = c;
        idx[1] = d.Gp + c;
        idx[2] = idx[3] = -1;
        if (bus_gen_has_ramp(d, c)) idx[2] = -2 - (c + d.g1);  // ramp partner: record 2 = (r_u, r_z, r_l, r_rho)
    } else {
        const int e = 2 * d.Gp + c;
        const int off = (kind == END_FROM) ? 0 : 2;
        idx[0] = e + off * Lp;                          // p
        idx[1] = e + (off + 1) * Lp;                    // q
        idx[2] = e + ((kind == END_FROM) ? 4 : 5) * Lp; // w
        idx[3] = e + ((kind == END_FROM) ? 6 : 7) * Lp; // theta
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int i = idx[m];
        if (i >= 0) {
            r[4 * m] = d.u[i];
            r[4 * m + 1] = d.z[i];
            r[4 * m + 2] = d.l[i];
            r[4 * m + 3] = d.rho[i];
        } else if (i <= -2) {
            const int ri = -2 - i;
            r[4 * m] = d.r_u[ri];
            r[4 * m + 1] = d.r_z[ri];
            r[4 * m + 2] = d.r_l[ri];
            r[4 * m + 3] = d.r_rho[ri];
        } else {
            r[4 * m] = r[4 * m + 1] = r[4 * m + 2] = 0.0;
            r[4 * m + 3] = 1.0;
        }
    }
}

// The rho-only part of the bus update, one lane per bus, in the order k_bus adds things up: rerun whenever rho changes
// (upload of rho, init_solution).  busc: BUSC_N planes of B doubles.
__global__ void k_bus_const(Dev d, double* busc, double baseMVA)
{
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= d.B) return;
    const int Lp = d.Lp, Gp = d.Gp, lb = 2 * d.Gp;
    const double* rho = d.rho;
    double inv_rhosum_p = 0.0, inv_rhosum_q = 0.0, rhosum_w = 0.0, rhosum_t = 0.0;
    for (int k = d.FrStart[I]; k < d.FrStart[I + 1]; k++) {
        const int e = lb + d.FrIdx[k];
        inv_rhosum_p += 1.0 / rho[e];
        inv_rhosum_q += 1.0 / rho[e + Lp];
        rhosum_w += rho[e + 4 * Lp];
        rhosum_t += rho[e + 6 * Lp];
    }
    for (int k = d.ToStart[I]; k < d.ToStart[I + 1]; k++) {
        const int e = lb + d.ToIdx[k];
        inv_rhosum_p += 1.0 / rho[e + 2 * Lp];
        inv_rhosum_q += 1.0 / rho[e + 3 * Lp];
        rhosum_w += rho[e + 5 * Lp];
        rhosum_t += rho[e + 7 * Lp];
    }
    double inv_rhosum_pg = 0.0, inv_rhosum_qg = 0.0;
    for (int k = d.GenStart[I]; k < d.GenStart[I + 1]; k++) {
        const int pg = d.GenIdx[k], qg = Gp + pg;
        if (bus_gen_has_ramp(d, pg))
            inv_rhosum_pg += 1.0 / (rho[pg] + d.r_rho[pg + d.g1]);
        else
            inv_rhosum_pg += 1.0 / rho[pg];
        inv_rhosum_qg += 1.0 / rho[qg];
    }
    const double ysr = d.YshR[I], ysi = d.YshI[I];
    const double A11 = (inv_rhosum_pg + inv_rhosum_p) + (ysr * ysr / rhosum_w);
    const double A12 = -ysr * (ysi / rhosum_w);
    const double A21 = A12;
    const double A22 = (inv_rhosum_qg + inv_rhosum_q) + (ysi * ysi / rhosum_w);
    const double r21 = A21 / A11;
    busc[BUSC_A11 * d.B + I] = A11;
    busc[BUSC_A12 * d.B + I] = A12;
    busc[BUSC_R21 * d.B + I] = r21;
    busc[BUSC_PIV * d.B + I] = A22 - r21 * A12;
    busc[BUSC_RW * d.B + I] = rhosum_w;
    busc[BUSC_RT * d.B + I] = rhosum_t;
    busc[BUSC_PD * d.B + I] = d.Pd[I] / baseMVA;   // (the quotients the bus kernels form: same operands, same bits)
    busc[BUSC_QD * d.B + I] = d.Qd[I] / baseMVA;
    busc[BUSC_YSR * d.B + I] = ysr;
    busc[BUSC_YSI * d.B + I] = ysi;
}

__device__ __forceinline__ void block_reduce4(double (&s)[4]);   // (defined with the elementwise kernels below)

// inner++, norms, stop test (admm_two_level.jl:45,60-62 evaluated on the device) from the four sums of squares
__device__ __forceinline__ void ctrl_advance(Ctrl* c, const double (&s)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c->sums[j] = s[j];
        c->res[j] = sqrt(s[j]);
    }
    const int inner = c->inner + 1;
    c->inner = inner;
    c->pending = 0;
    const bool test = (c->check_every <= 1) || (inner % c->check_every == 0);
    if ((test && c->res[0] <= c->eps_pri) || inner >= c->max_inner) c->done = 1;
}

// the same from the four NORMS (multi-period model: per-period norms, maximum over the periods -- mpacopf_admm_update_residual_cpu.jl:1-55)
__device__ __forceinline__ void ctrl_advance_norms(Ctrl* c, const double (&best)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c->sums[j] = best[j] * best[j];
        c->res[j] = best[j];
    }
    const int inner = c->inner + 1;
    c->inner = inner;
    c->pending = 0;
    const bool test = (c->check_every <= 1) || (inner % c->check_every == 0);
    if ((test && c->res[0] <= c->eps_pri) || inner >= c->max_inner) c->done = 1;
}

// index of the v entries a component of a bus receives (generator: pg, qg; branch end: p, q, w, theta of that end)
__device__ __forceinline__ void bus_end_index(const Dev& d, int code, int (&idx)[4])
{
    const int kind = (code >> 28) & 3, c = code & END_INDEX, Lp = d.Lp;
    if (kind == END_GEN) {
        idx[0] = c;
        idx[1] = d.Gp + c;
        idx[2] = idx[3] = -1;
    } else {
        const int e = 2 * d.Gp + c;
        const int off = (kind == END_FROM) ? 0 : 2;
        idx[0] = e + off * Lp;
        idx[1] = e + (off + 1) * Lp;
        idx[2] = e + ((kind == END_FROM) ? 4 : 5) * Lp;
        idx[3] = e + ((kind == END_FROM) ? 6 : 7) * Lp;
    }
}

// red != nullptr (the speculative loop): the launch carries one extra block that sums the per-block residual terms the
// preceding branch launch's prologue left in partials_br.  One GPU: it advances the loop control itself (what
// k_final_reduce does).  A partition with peers: it appends the four partial sums to every peer's halo message; the loop
// control is advanced by k_halo_unpack once the peers' sums have arrived.
// REC = false: every lane gathers its component's (u, z, l, rho) from the solution planes (step-wise path, multi-period
// model).  REC = true (the fused loop): the branch launch has pushed, for every component, the 8 doubles the bus update
// needs into d.rec at the slot this lane owns (see Dev::rec) -- the launch starts with ONE round of loads whose addresses
// all follow from the thread id, instead of an index load followed by 16 dependent gathers per lane spread over 16
// cache lines; same terms, same order, same bits.
// Parameter order: what a wavefront needs for its FIRST loads -- loop control, record / slot tables, the bus constants --
// leads, inside the 16 SGPRs the firmware preloads at wavefront launch (Makefile KERNARG_PRELOAD); the Dev block by value
// follows and is read from the kernarg segment while those loads are in flight.  hot_* repeat d.rec, d.EndCount, d.SlotCode, d.busc.
// MPG = true (multi-period model on the fused loop, with REC): the real power of a generator of period t < T - 1 is also tied to the
// ramp element of the next period (bus_gen_has_ramp); that element's lane has put its consensus term and its rho into fields 6
// and 7 of the generator's record (gen_ramp_lane_zl).
template <bool REC, bool MPG = false>
__global__ void __launch_bounds__(BUS_BLOCK) k_bus_group_t(const Ctrl* ctrl, Ctrl* red, const double* __restrict__ hot_rec,
                                                            const int* __restrict__ hot_count, const int* __restrict__ hot_slot,
                                                            const double* __restrict__ hot_busc, int hot_B, Dev d, double baseMVA)
{
    // Terms a lane hands to its group leader.  The leader adds them in the reference's order -- generators, then -Pd, then the
    // branch ends -- WITHOUT looking at what kind of component a slot holds: a slot contributes a neutral element to the
    // sums it does not belong to (-0.0 where the leader adds, +0.0 where it subtracts: x + (-0) = x and x - (+0) = x for
    // every x, signed zeros included), and so do the slots past the bus's last component.  Same effective operations in
    // the same order as the component-by-component loop, hence the same bits; no branch and no dependent LDS read in the
    // leader's chain (3.3 k -> ~1.5 k clocks per block, scripts/bus_stamps.py).
    constexpr int T_W = 0, T_T = 1, T_PG = 2, T_QG = 3, T_PE = 4, T_QE = 5, NTERM = 6;
    __shared__ double term[BUS_BLOCK / BUS_W][NTERM][BUS_W];
    __shared__ double res[BUS_BLOCK / BUS_W][4];
    // (the reduction block is block 0: dispatched first, it runs beside the bus blocks instead of behind the last of them)
    if (red != nullptr && blockIdx.x == 0) {   // the reduction block (whole block takes this path)
        // (one GPU: `pending` says whether the preceding branch launch made a z / l pass.  A partition with peers passes red =
        //  nullptr instead when it did not: there the flag is rewritten by k_halo_unpack on another stream while the next
        //  interior branch launch is already setting it again)
        if (red->done || (d.nranks <= 1 && !red->pending)) return;
        if (d.T > 1) {
            // multi-period model: per period the sums of the blocks that hold its lanes (branch blocks, then generator blocks, whose
            // terms include the ramp elements), norm = sqrt, model norm = max over the periods (k_final_reduce_mp's definition)
            double best[4] = {0.0, 0.0, 0.0, 0.0};
            for (int t = 0; t < d.T; t++) {
                double s4[4] = {0.0, 0.0, 0.0, 0.0};
                const int b0 = (t * d.l1) / BR_BLOCK, b1 = ((t + 1) * d.l1 - 1) / BR_BLOCK;
                for (int bq = b0 + (int)threadIdx.x; bq <= b1; bq += BUS_BLOCK) {
                    const int k = t - (bq * BR_BLOCK) / d.l1;
#pragma unroll
                    for (int j = 0; j < 4; j++) s4[j] += d.partials_br[((size_t)bq * d.part_k + k) * 4 + j];
                }
                const int g0 = (t * d.g1) / BR_BLOCK, g1 = ((t + 1) * d.g1 - 1) / BR_BLOCK;
                for (int gq = g0 + (int)threadIdx.x; gq <= g1; gq += BUS_BLOCK) {
                    const int k = t - (gq * BR_BLOCK) / d.g1;
#pragma unroll
                    for (int j = 0; j < 4; j++) s4[j] += d.partials_br[((size_t)(d.gen_part0 + gq) * d.part_k + k) * 4 + j];
                }
                block_reduce4(s4);
                if (threadIdx.x == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) best[j] = fmax(best[j], sqrt(s4[j]));
                }
                __syncthreads();   // block_reduce4's scratch is reused by the next period
            }
            if (threadIdx.x == 0) ctrl_advance_norms(red, best);
            return;
        }
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int bq = threadIdx.x; bq < d.npartials_br; bq += BUS_BLOCK) {
#pragma unroll
            for (int j = 0; j < 4; j++) s4[j] += d.partials_br[bq * 4 + j];
        }
        block_reduce4(s4);
        if (threadIdx.x == 0) {
            if (d.nranks > 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) d.halo_local[j] = s4[j];
                for (int p = 0; p < d.nranks; p++) {
                    if (p == d.rank) continue;
#pragma unroll
                    for (int j = 0; j < 4; j++) d.halo_send[d.halo_off[p] + j] = s4[j];
                }
            } else
                ctrl_advance(red, s4);
        }
        return;
    }
    if (loop_done(ctrl)) return;
    const int grp = threadIdx.x / BUS_W, lane = threadIdx.x % BUS_W;
    const int bus_block = (int)blockIdx.x - (red != nullptr ? 1 : 0);
    const int I = bus_block * (BUS_BLOCK / BUS_W) + grp;
#if defined(EXA_PROFILE_PHASES)
    unsigned long long tb0 = __builtin_amdgcn_s_memtime(), tb1 = 0, tb2 = 0, tb3 = 0, tb4 = 0;
#endif
    if (I >= hot_B) return;   // whole groups leave together
    const int n = hot_count[I];
    const int code0 = hot_slot[I * BUS_W + lane];
    const int e0 = (n > BUS_W) ? d.EndStart[I] : 0;   // (only the rare large buses need the CSR offset)
    // REC: the first chunk's records, fetched with everything else (a lane past the bus's component count reads its own,
    // unused, slot: no branch in front of the loads)
    double r[BUS_REC];
    if (REC) {
        const double* q = hot_rec + 8 * (size_t)(I * BUS_W + lane);
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = q[k];
    }
#if defined(EXA_PROFILE_PHASES)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb1) : "v"(n) : "memory");
#endif

    // leader state (meaningful in lane 0 only)
    double common_wi = 0.0, common_ti = 0.0, rhs1 = 0.0, rhs2 = 0.0;
    bool load_subtracted = false;
    const double pd = hot_busc[BUSC_PD * hot_B + I], qd = hot_busc[BUSC_QD * hot_B + I];
    const double ysr = hot_busc[BUSC_YSR * hot_B + I], ysi = hot_busc[BUSC_YSI * hot_B + I];
    // the bus's rho-only constants (k_bus_const), fetched with everything else ahead of the wavefront fences
    const double A11 = hot_busc[BUSC_A11 * hot_B + I], A12 = hot_busc[BUSC_A12 * hot_B + I], r21 = hot_busc[BUSC_R21 * hot_B + I];
    const double piv = hot_busc[BUSC_PIV * hot_B + I], rhosum_w = hot_busc[BUSC_RW * hot_B + I], rhosum_t = hot_busc[BUSC_RT * hot_B + I];

    int idx[4], kind = -1, code = 0;
    for (int c0 = 0; c0 < n; c0 += BUS_W) {
        const int m = min(BUS_W, n - c0);
        if (lane < m) {
            code = (c0 == 0) ? code0 : d.EndCode[e0 + c0 + lane];
            kind = (code >> 28) & 3;
            const bool rec_lane = REC;
            const bool ramp_gen = MPG && kind == END_GEN && bus_gen_has_ramp(d, code & END_INDEX);
            if (rec_lane) {
                if (c0 > 0) {
                    const double* q = d.rec + 8 * (size_t)(BUS_W * d.B + e0 + c0 + lane);
#pragma unroll
                    for (int k = 0; k < 8; k++) r[k] = q[k];
                }
                bus_end_index(d, code, idx);
            } else
                bus_load_end(d, code, r, idx);
#if defined(EXA_PROFILE_PHASES)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb2) : "v"(r[0]) : "memory");
#endif
            double tp, tq, tw = -0.0, tt = -0.0;
            if (rec_lane) {   // record: (u+z, l, rho) of p / pg, the same of q / qg, the w term, the theta term
                tp = ramp_gen ? ((r[1] + r[2] * r[0]) + r[6]) / (r[2] + r[7]) : r[0] + (r[1] / r[2]);
                tq = r[3] + (r[4] / r[5]);
                if (kind != END_GEN) {
                    tw = r[6];
                    tt = r[7];
                }
            } else {
                if (idx[2] <= -2)  // generator with a ramp partner (mpacopf_bus_kernel_cpu.jl:53-62)
                    tp = ((r[2] + r[3] * (r[0] + r[1])) + (r[10] + r[11] * (r[8] + r[9]))) / (r[3] + r[11]);
                else
                    tp = (r[0] + r[1]) + (r[2] / r[3]);
                tq = (r[4] + r[5]) + (r[6] / r[7]);
                if (kind != END_GEN) {
                    tw = r[10] + r[11] * (r[8] + r[9]);
                    tt = r[14] + r[15] * (r[12] + r[13]);
                }
            }
            const bool gen = (kind == END_GEN);
            term[grp][T_PG][lane] = gen ? tp : -0.0;
            term[grp][T_QG][lane] = gen ? tq : -0.0;
            term[grp][T_PE][lane] = gen ? 0.0 : tp;
            term[grp][T_QE][lane] = gen ? 0.0 : tq;
            term[grp][T_W][lane] = tw;
            term[grp][T_T][lane] = tt;
        } else {   // no component in this slot: neutral elements
            term[grp][T_PG][lane] = -0.0;
            term[grp][T_QG][lane] = -0.0;
            term[grp][T_PE][lane] = 0.0;
            term[grp][T_QE][lane] = 0.0;
            term[grp][T_W][lane] = -0.0;
            term[grp][T_T][lane] = -0.0;
        }
        // does this chunk hold a branch end?  (the load is subtracted once, in front of the first one)
        const unsigned long long ends = __ballot(lane < m && kind != END_GEN);
        const bool has_end = ((ends >> (8 * ((threadIdx.x & 63) / BUS_W))) & 0xffull) != 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < BUS_W; k++) {
                rhs1 += term[grp][T_PG][k];
                rhs2 += term[grp][T_QG][k];
            }
            if (has_end && !load_subtracted) {
                rhs1 -= pd;
                rhs2 -= qd;
                load_subtracted = true;
            }
#pragma unroll
            for (int k = 0; k < BUS_W; k++) {
                common_wi += term[grp][T_W][k];
                common_ti += term[grp][T_T][k];
                rhs1 -= term[grp][T_PE][k];
                rhs2 -= term[grp][T_QE][k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        if (!load_subtracted) {
            rhs1 -= pd;
            rhs2 -= qd;
        }
        common_wi /= rhosum_w;
        rhs1 -= ysr * common_wi;
        rhs2 += ysi * common_wi;
        const double mu2 = (rhs2 - r21 * rhs1) / piv;
        const double mu1 = (rhs1 - A12 * mu2) / A11;
        res[grp][0] = mu1;
        res[grp][1] = mu2;
        res[grp][2] = common_wi + ((ysr * mu1 - ysi * mu2) / rhosum_w);
        res[grp][3] = common_ti / rhosum_t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double mu1 = res[grp][0], mu2 = res[grp][1], wi = res[grp][2], ti = res[grp][3];
#if defined(EXA_PROFILE_PHASES)
    tb3 = __builtin_amdgcn_s_memtime();
#endif

    for (int c0 = 0; c0 < n; c0 += BUS_W) {
        const int m = min(BUS_W, n - c0);
        if (lane < m) {
            if (n > BUS_W) {  // registers only hold the last chunk: fetch this one again
                code = (c0 == 0) ? code0 : d.EndCode[e0 + c0 + lane];
                kind = (code >> 28) & 3;
                if (REC) {
                    const double* q = d.rec + 8 * (size_t)((c0 == 0) ? (I * BUS_W + lane) : (BUS_W * d.B + e0 + c0 + lane));
#pragma unroll
                    for (int k = 0; k < 8; k++) r[k] = q[k];
                    bus_end_index(d, code, idx);
                } else
                    bus_load_end(d, code, r, idx);
            }
            const bool rec_lane = REC;
            // (u+z), l, rho of the component's p (pg) and q (qg) entries
            const double uzp = rec_lane ? r[0] : (r[0] + r[1]), lp = rec_lane ? r[1] : r[2], rhop = rec_lane ? r[2] : r[3];
            const double uzq = rec_lane ? r[3] : (r[4] + r[5]), lq = rec_lane ? r[4] : r[6], rhoq = rec_lane ? r[5] : r[7];
            if (kind == END_GEN) {
                if (!rec_lane && idx[2] <= -2)
                    d.v[idx[0]] = ((r[2] + r[3] * (r[0] + r[1])) + (r[10] + r[11] * (r[8] + r[9])) - mu1) / (r[3] + r[11]);
                else if (MPG && bus_gen_has_ramp(d, code & END_INDEX))
                    d.v[idx[0]] = ((r[1] + r[2] * r[0]) + r[6] - mu1) / (r[2] + r[7]);
                else
                    d.v[idx[0]] = uzp + (lp - mu1) / rhop;
                d.v[idx[1]] = uzq + (lq - mu2) / rhoq;
            } else {
                const double vp = uzp + (lp + mu1) / rhop;
                const double vq = uzq + (lq + mu2) / rhoq;
                d.v[idx[0]] = vp;
                d.v[idx[1]] = vq;
                d.v[idx[2]] = wi;
                d.v[idx[3]] = ti;
                if ((code & END_CUT) && d.halo_pos != nullptr) {   // this end's new v entries are the peer's missing half of the branch: into its message
                                                                    // (attached partitions only; the step-wise twin gathers them with exaadmm_halo_pack)
                    double* q = d.halo_send + d.halo_pos[code & END_INDEX];
                    q[0] = vp;
                    q[1] = vq;
                    q[2] = wi;
                    q[3] = ti;
                }
            }
        }
    }
#if defined(EXA_PROFILE_PHASES)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb4) : : "memory");
    if (d.st_tl && threadIdx.x == 0 && bus_block < 256) {   // per block: entry, CSR landed, gathers landed, leader done, stores landed
        unsigned long long* q = d.st_tl + (size_t)bus_block * 8;
        q[0] = tb0; q[1] = tb1; q[2] = tb2; q[3] = tb3; q[4] = tb4;
    }
#endif
}

// ---------------------------------------------------------------- resident loop
// The inner loop of grids whose wavefronts all fit the chip at once (one wavefront per SIMD: branch + generator
// wavefronts + 1 <= 4 x CUs - 8; 13 659 buses: 385 of 1 024).  ALL wavefronts stay resident for a batch of iterations;
// nothing is launched between them (exaadmm_hip.hip enqueue_resident; EXAADMM_NO_RESIDENT=1 selects the two-launch loop).  What the bus
// launch did for everybody, the last lane to ARRIVE at a bus does for that bus:
//   * a lane that has finished its solve of iteration j pushes the four sum terms of its two ends (tp, tq, tw, tt as the
//     bus kernels form them) to its slots of terms[j & 1], fences, and counts itself in at both buses (arrive[I] += 1);
//   * the lane whose arrival completes bus I (arrive[I] == components x j) reads the bus's terms, makes the bus update in
//     the reference's order (bus_from_rec4: same bits as k_bus) and publishes (mu1, mu2, w, theta) in busres[j & 1][I],
//     then ready[I] = j (release);
//   * a lane starts iteration j + 1 when both its buses are ready for j (acquire), forms the v entries of its own
//     elements, runs the z / l pass and the solve exactly as k_branch_t<false, true> does.
// The launch's slowest branch is the last to arrive at its own buses: it goes from one solve to the next without waiting
// for anybody (scripts/microbench/flag_chain.hip: ~1.5 us per hand-over), where the two-launch loop puts two kernel
// boundaries, the bus launch and two rounds of loads: 57.8 -> 52 us per iteration at 13 659 buses, whole solve 7.5 k -> 9.5 k it/s.
// Same arithmetic per lane, same order in every sum: bitwise the two-launch loop (tests/test_gpu_resident.py).
// Every wait is capped: a wavefront that cannot become resident (the GPU shared with other work) ends the launch with
// rs.fail set; the host then restores the iterate the call started from and repeats the loop with two launches per iteration
// (exaadmm_hip.hip resident_downgrade) -- the call does not fail and does not hang.
// What wavefronts hand to each other -- sum terms, bus results, counters, flags -- is written and read with agent-scope
// (sc1) accesses: they go to the coherence point themselves, so no L2 write-back / invalidate (agent-scope fence) is needed
// inside the loop; s_waitcnt vmcnt(0) between the data and the flag / counter that announces them orders them.  With agent-scope fences the loop ran at 84 us per
// iteration at 13 659 buses: ~1 900 L2 write-backs per iteration queue up behind each other.
// everything this wavefront has stored is performed (for sc1 stores: at the coherence point) before what follows is issued.
// (A workgroup-scope fence compiles to NOTHING on gfx950 -- checked -- and left the flag free to overtake the data.)
__device__ __forceinline__ void resident_order_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct ResidentSync {
    int* arrive;       // [B] arrivals so far (monotonic over the launch)
    int* ready;        // [B] last iteration whose bus update is published
    double* terms[2];  // [slots][4], slot numbering of Dev::rec
    double* busres[2]; // [B][4]
    double* partials[2];   // [work blocks][4] residual terms of the z / l pass of index parity
    int* contrib;      // contributions to the residual terms so far (monotonic over the launch)
    int* decided;      // decisions published so far
    int* stop;         // the last decision said stop
    int* fail;
    int* last3;        // [Lp] (TILES = 1) (resident iteration in which the branch last needed >= 3 solver trips) << 4 | those trips; 0: never.  Advisory:
                       //  k_heavy_select reads it in front of a later launch (a straggler that rests for two iterations is still a straggler)
    int stamp0;        // resident iterations of this handle before this launch (its index i is iteration stamp0 + i + 1 of that count)
};

__device__ __forceinline__ void bus_from_rec4(const Dev& d, const double* __restrict__ rec4, int I, int cnt, double& mu1, double& mu2,
                                              double& wi, double& ti)
{
    const int n = cnt & 0xffff, ngen = cnt >> 16;
    const int B = d.B;
    const double* const c = d.busc + I;
    const double A11 = c[BUSC_A11 * B], A12 = c[BUSC_A12 * B], r21 = c[BUSC_R21 * B], piv = c[BUSC_PIV * B];
    const double rhosum_w = c[BUSC_RW * B], rhosum_t = c[BUSC_RT * B];
    const double pd = c[BUSC_PD * B], qd = c[BUSC_QD * B], ysr = c[BUSC_YSR * B], ysi = c[BUSC_YSI * B];
    double t[BUS_W][4];
#pragma unroll
    for (int k = 0; k < BUS_W; k++) {
        if (k < n) {
            const double* q = rec4 + 4 * (size_t)(I * BUS_W + k);
#pragma unroll
            for (int m = 0; m < 4; m++) t[k][m] = ld_agent(q + m);
        }
    }
    double rhs1 = 0.0, rhs2 = 0.0, cw = 0.0, ct = 0.0;
    bool sub = false;
#pragma unroll
    for (int k = 0; k < BUS_W; k++) {
        if (k < n) {
            if (k < ngen) {
                rhs1 += t[k][0];
                rhs2 += t[k][1];
            } else {
                if (!sub) {
                    rhs1 -= pd;
                    rhs2 -= qd;
                    sub = true;
                }
                cw += t[k][2];
                ct += t[k][3];
                rhs1 -= t[k][0];
                rhs2 -= t[k][1];
            }
        }
    }
    if (n > BUS_W) {   // the rare large bus
        const int e0 = d.EndStart[I];
        for (int k = BUS_W; k < n; k++) {
            const double* q = rec4 + 4 * (size_t)(BUS_W * B + e0 + k);
            const double a = ld_agent(q), b = ld_agent(q + 1), w = ld_agent(q + 2), th = ld_agent(q + 3);
            if (k < ngen) {
                rhs1 += a;
                rhs2 += b;
            } else {
                if (!sub) {
                    rhs1 -= pd;
                    rhs2 -= qd;
                    sub = true;
                }
                cw += w;
                ct += th;
                rhs1 -= a;
                rhs2 -= b;
            }
        }
    }
    if (!sub) {
        rhs1 -= pd;
        rhs2 -= qd;
    }
    cw /= rhosum_w;
    rhs1 -= ysr * cw;
    rhs2 += ysi * cw;
    mu2 = (rhs2 - r21 * rhs1) / piv;
    mu1 = (rhs1 - A12 * mu2) / A11;
    wi = cw + ((ysr * mu1 - ysi * mu2) / rhosum_w);
    ti = ct / rhosum_t;
}

__device__ __forceinline__ bool resident_wait(const int* flag, int want, int* fail, int code = 1)
{
    // Polls are RELAXED agent-scope loads (an acquire poll invalidates the XCD's L2 for everybody, hundreds of waiting
    // wavefronts at a time: the working wavefronts slowed down 2.6 x).  There is NO fence behind the poll, only a compiler
    // barrier: the invariant of the loop is that EVERY word one wavefront hands to another -- terms, bus results, residual
    // terms, counters, flags -- is written with st_agent / an agent-scope atomic and read with ld_agent, i.e. performed at the
    // device's coherence point and never served from a per-XCD L2 line; the writer's s_waitcnt vmcnt(0) between data and
    // flag makes the flag the later event there.  A plain load of hand-over data would be a bug (stale L2 across XCDs).
    // The wait is capped, and a wavefront that sees another one's failure gives up at once (one failed wait would otherwise
    // cost every waiter its own cap): the launch ends with rs.fail set and the host repeats the batch with the two-launch loop.
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        ++spins;
        if (spins > (1 << 20) || ((spins & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            int expected = 0;   // (the FIRST failure's code stays: 1 = a bus of this GPU, 6 = the control wavefront's wait for the residual terms, 7 = its decision)
            (void)__hip_atomic_compare_exchange_strong(fail, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(24);
    }
    asm volatile("" ::: "memory");
    return true;
}

// Two flags with ONE round trip per poll (both loads in flight together): a branch lane's two buses.  Same cap, same failure code as two
// waits one after the other (which cost a lone lane a second round trip of ~0.5 us although both flags were up long ago).
__device__ __forceinline__ bool resident_wait2(const int* fa, const int* fb, int want, int* fail, int code = 1)
{
    int spins = 0;
    for (;;) {
        const int a = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a >= want && b >= want) break;
        ++spins;
        if (spins > (1 << 20) || ((spins & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            int expected = 0;
            (void)__hip_atomic_compare_exchange_strong(fail, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(24);
    }
    asm volatile("" ::: "memory");
    return true;
}

// after the lane's terms of iteration j (1-based) are pushed and fenced.  Returns true when this lane was the last to arrive and has
// updated the bus: res = what it published (mu1, mu2, w, theta) -- a lane that needs the result itself need not read it back.
__device__ __forceinline__ bool resident_arrive_keep(const Dev& d, const ResidentSync& rs, int I, int cnt, int j, double (&res)[4],
                                                     unsigned long long* stamps = nullptr)   // stamps: diagnostic build, 3 clocks (counter back, update computed, stores acknowledged)
{
    const int n = cnt & 0xffff;
    const int old = __hip_atomic_fetch_add(rs.arrive + I, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 != n * j) return false;
    asm volatile("" ::: "memory");
#if defined(EXA_PROFILE_PHASES)
    if (stamps) stamps[0] = __builtin_amdgcn_s_memtime();
#endif
    double mu1, mu2, wi, ti;
    bus_from_rec4(d, rs.terms[j & 1], I, cnt, mu1, mu2, wi, ti);
#if defined(EXA_PROFILE_PHASES)
    if (stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamps[1]) : "v"(mu1), "v"(wi), "v"(ti) : "memory");
#endif
    double* q = rs.busres[j & 1] + 4 * (size_t)I;
    st_agent(q, mu1);
    st_agent(q + 1, mu2);
    st_agent(q + 2, wi);
    st_agent(q + 3, ti);
    resident_order_stores();
#if defined(EXA_PROFILE_PHASES)
    if (stamps) stamps[2] = __builtin_amdgcn_s_memtime();
#endif
    __hip_atomic_store(rs.ready + I, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    res[0] = mu1;
    res[1] = mu2;
    res[2] = wi;
    res[3] = ti;
    return true;
}
__device__ __forceinline__ void resident_arrive(const Dev& d, const ResidentSync& rs, int I, int cnt, int j)
{
    double res[4];
    (void)resident_arrive_keep(d, rs, I, cnt, j, res);
}

// Loop control of the resident launch.  Index i of the launch = iteration ctrl->base + i + 1.  The z / l pass of iteration m - 1
// runs in the prologue of index i (m = base + i + 1) unless m is the loop's first iteration; its residual terms decide whether
// the loop stops AFTER iteration m - 1.  The decision is made by the control wavefront (the launch's last block) once every
// other wavefront has contributed its terms of that pass (contrib counter), in the reduction order of the bus launch's
// reduction block (same bits), and published through `decided`; a wavefront reads it before the z / l pass of the NEXT index
// overwrites the iterate that would be final -- exactly the one-launch speculation of the two-launch loop.
// (first_has_zl is read from the control block ONCE, at kernel entry: the control wavefront rewrites that cache line with
//  plain stores while the launch runs)
__device__ __forceinline__ int resident_passes(bool first_has_zl, int i) { return (i + 1) - (first_has_zl ? 0 : 1); }

__device__ __forceinline__ void resident_contribute(const ResidentSync& rs, int parity, int part, const double (&zs)[4])
{
    // (lane 0 of a wavefront, after the wave reduction of its terms; part = the tile's slot in the reduction order: its block
    //  index in the two-launch loop)
    double* q = rs.partials[parity] + 4 * (size_t)part;
#pragma unroll
    for (int j = 0; j < 4; j++) st_agent(q + j, zs[j]);
    resident_order_stores();
    __hip_atomic_fetch_add(rs.contrib, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The decision word: `decided` = number of decisions published so far, bit 30 set by the decision that said stop (ONE word: a reader
// needs one round trip, the control wavefront no acknowledgement between two stores; rs.stop is no longer read inside the launch).
constexpr int RESIDENT_STOP_BIT = 1 << 30;
// returns false when the loop has stopped (or a wait failed)
__device__ __forceinline__ bool resident_go_on(const ResidentSync& rs, int passes_needed)
{
    if (passes_needed <= 0) return true;
    int spins = 0, v;
    while ((v = __hip_atomic_load(rs.decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < passes_needed) {   // (a word with the stop bit is >= every count)
        ++spins;
        if (spins > (1 << 20) || ((spins & 255) == 0 && __hip_atomic_load(rs.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            int expected = 0;
            (void)__hip_atomic_compare_exchange_strong(rs.fail, &expected, 7, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(24);
    }
    asm volatile("" ::: "memory");
    return (v & RESIDENT_STOP_BIT) == 0;
}

// Top of an index of a HEAVY wavefront: the decision and the ready flags of its two buses asked for in ONE round of loads (its lone
// arriving lanes are the launch's critical path: every round trip there is ~0.5 us of every iteration).  fa / fb = nullptr: the
// wavefront made that bus's update itself (resident_arrive_keep) and waits for nobody.  Same order of precedence as resident_go_on
// followed by the two waits: a published stop ends the loop whatever the flags say.
__device__ __forceinline__ bool resident_top_wait(const ResidentSync& rs, int passes_needed, const int* fa, const int* fb, int want)
{
    int spins = 0;
    for (;;) {
        // (passes_needed = 0: no decision is waited for, but a published stop still ends the wait -- a wavefront that returned on it has not
        //  arrived at the buses this one is waiting for)
        // (... looked for in every 16th poll only: ~200 waiting heavy wavefronts polling the decision word in step with their flags slowed
        //  the decisions themselves down -- 22.6 k -> 21.1 k it/s at 13 659 buses)
        const bool poll = (passes_needed > 0) || ((fa != nullptr || fb != nullptr) && (spins & 15) == 15);
        const int v = poll ? __hip_atomic_load(rs.decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int a = fa ? __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
        const int b = fb ? __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
        const bool dec = (passes_needed <= 0) || (v >= passes_needed);
        if ((v & RESIDENT_STOP_BIT) != 0) return false;
        if (dec && a >= want && b >= want) break;
        ++spins;
        if (spins > (1 << 20) || ((spins & 255) == 0 && __hip_atomic_load(rs.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            int expected = 0;
            (void)__hip_atomic_compare_exchange_strong(rs.fail, &expected, dec ? 1 : 7, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(24);
    }
    asm volatile("" ::: "memory");
    return true;
}

// TILES = 2: grids with more tiles of 64 branches than the chip has SIMDs (up to twice as many: ~120 000 branches).  A resident
// wavefront then owns up to two tiles and runs, per iteration, the body once per tile (the per-lane state lives in memory between
// iterations anyway).  Which tiles share a wavefront is decided per launch by k_tile_keys / k_tile_assign from the solver-trip counters
// of the last iterations: the heaviest tiles -- the launch's slowest branch sets the pace of every iteration, and the set of
// heavy branches is persistent -- get a wavefront of their own, the others are paired heavy with light, heavier first.
// tiles[2 w], tiles[2 w + 1] = the tiles of branch wavefront w (-1: none); generator wavefront v handles generator tiles v and
// v + wg.  Waits only ever refer to earlier (iteration, position) pairs, so the order inside a wavefront cannot deadlock.
// Partial residual sums are indexed by TILE (the block index of the two-launch loop): the control wavefront adds them in the
// same order as before -- same bits.
struct ResidentTiles {
    const int* tiles;
    int nb, ng, wb, wg;   // branch tiles, generator tiles, branch wavefronts, generator wavefronts
};

// HEAVY WAVEFRONTS (TILES = 1: grids that leave SIMDs idle -- 13 659 buses use 385 of 1 024).  What sets the pace of an iteration is
// its slowest branch: six solver trips, four or five of them with ONE active lane in its wavefront.  The branches that needed three
// or more trips in the last iterations (k_heavy_select, in front of the launch; the set is persistent) get a wavefront of their
// own, blocks [nwaves, nwaves + rh.n): all 64 lanes run THAT branch's solve as clones -- the same inputs, instructions and bits in
// every lane -- except that in the Cauchy search every lane evaluates its own candidate step length (tron_device.h cauchy_clones:
// 3.8 model evaluations one after the other become one).  The branch's lane in its tile still makes the z / l pass and the
// residual contribution (the sums keep their order) and stores v, z, l; it neither solves nor arrives.  The heavy wavefront keeps
// z, l and the multipliers of its branch in registers (it forms the same z / l values from the same inputs), solves, stores u
// (agent scope: the tile's lane reads it for the next z / l pass, after its buses are ready -- which the heavy wavefront's
// arrival precedes), pushes the bus terms and arrives at both buses.  Same arithmetic per branch, whoever solves it: bitwise the
// two-launch loop (tests/test_gpu_resident.py).  This code shares no value with the solver loop of the ordinary lanes: inlined
// into that loop the side-by-side search cost every trip of every lane 8 - 21 % (profiles/archive_r03_r04/r04_lone_search.txt).
struct ResidentHeavy {
    const int* list;   // [n] storage slot of the branch of heavy wavefront w, -1: none
    int* flag;         // [Lp] w + 1: the branch is solved by heavy wavefront w in this launch, 0: by its lane (bit 20: adopted at the end of the last index)
    const double* init;   // [n][16] z, l of the heavy branches in front of the launch (k_heavy_select)
    // ADOPTION (adopt_thr > 0): a branch lane whose solve took >= adopt_thr trips hands its branch over to one of the launch's FREE heavy
    // wavefronts (those behind the cnt[0] chosen in front of the launch), from the next index on: it takes a ticket (cnt[1]), writes the
    // branch's state -- u, z, l, multipliers, penalty: 27 doubles -- into adopt[w] with agent-scope stores and, behind their acknowledgement,
    // req[w] = (branch + 1) | (first index << 20); it marks flag[b] = (w + 1) | fresh (its own next index takes u from the planes it has just
    // written, every later one from the heavy wavefront's hand-over).  A free heavy wavefront polls req[w] until a request, a stop, or the
    // launch's last decision.  New stragglers emerge all the time (and a launch chose its heavy wavefronts before it began); a straggler that
    // stays with its tile paces the whole launch until the next launch.
    double* adopt;     // [n][32]
    int* req;          // [n]
    int* cnt;          // [0] heavy wavefronts chosen in front of the launch, [1] tickets taken so far
    int adopt_thr;
    double* u;         // [2][n][16] hand-over of a heavy branch's u (8 of the 16) to its tile's lane, by iteration parity -- lines that
                       //  only agent-scope accesses ever touch.  (NOT through the u plane: the tile's wavefront holds that line in its
                       //  XCD's L2 for its plain accesses to the neighbouring branches, and an agent-scope load of such a line is
                       //  served from there, stale -- seen, not assumed.)  No WORD is written by both wavefronts: one word dirty in
                       //  two L2s leaves the result to the order of the write-backs (seen: v of the last index).
    int n;
};

// RESIDENT LOOP OF A PARTITION WITH PEERS (PEERS = true; one process per GPU).  The loop keeps its form -- every
// wavefront resident, a bus updated by the last lane to arrive at it, a control wavefront that makes the stop test -- and what has to
// cross GPUs goes by DIRECT STORES over xGMI into the peer's mailbox (fine-grained device memory the peer has mapped for us), from
// inside the launch: no kernel boundary, no collective call, no host in the loop.
//   * a cut branch is solved on both of its owners (bit-identical replicas, as in the two-launch loop).  Its lane arrives at its LOCAL
//     bus only; what it is missing for its next iteration is the update of the bus at its other end, which the peer's lanes make.  The
//     twin lane on the peer waits for that bus anyway: as soon as it is ready the twin forwards the bus result (mu1, mu2, w, theta: 4
//     doubles) into our mailbox entry of that branch, and we do the same for it -- send first, then wait: no cycle;
//   * the control wavefronts exchange their four partial residual sums the same way and add them IN RANK ORDER (the order of
//     k_halo_unpack: the same bits and therefore the same decision in the same iteration on every rank).
// Mailbox words are SELF-VALIDATING (p2p_put / p2p_get below): 8 bytes = (half a double, sequence number) in one 64-bit access at
// system scope; a reader polls until the words carry the sequence number it waits for.  An entry is one 128-byte line: 8 words per
// iteration parity; a sender reuses a parity only after the receiver has consumed it (its next send depends on our answer to this one).
// Waits are capped like every wait of the loop; a failed wait ends the launch with rs.fail set, the host agrees on the outcome with
// its peers (exaadmm_hip.hip) and every rank repeats the call with the two-launch loop.  The ranks need not ARRIVE together: they meet in
// a kernel of their own in front of the launch (k_p2p_rendezvous, patient), so the caps in here only measure the loop, never a late host.
constexpr int P2P_MAX_RANKS = 16;
constexpr int P2P_ENTRY = 16;   // doubles per mailbox entry (128 bytes)
struct ResidentPeers {
    const int* cut_code;                        // [Lp] -1: both buses local; else (mailbox entry << 1) | (1: the TO end's bus is the peer's, 0: the FROM end's)
    const unsigned long long* twin;             // [ncut] address of the twin lane's entry in the PEER's mailbox
    double* box;                                // this rank's mailbox: ncut branch entries, then one entry per rank (residual sums)
    const unsigned long long* sums_at;          // [nranks] (device memory: lanes index it) address of OUR sums entry in rank q's mailbox
    int nranks, rank, ncut;
    int epoch;   // launches of this kind so far, the same on every rank: sequence numbers are (epoch << 12) | iteration -- unique across launches
    int delay_ticks;   // test build only (-DEXA_TEST_HOOKS, exaadmm_test_hook "p2p_delay_us"): every send waits this many 100 MHz ticks first
    long long patient_ticks;   // how long the host-issued kernels wait for a peer (P2P_PATIENT_TICKS; the test build's hook "p2p_patient_ms" shortens it)
};
// A mailbox WORD is 8 bytes: (32 bits of payload, 32-bit sequence number), written and read as ONE 64-bit access at system scope
// (sc0 sc1: performed at the memory side, not in a cache of either GPU).  An aligned 64-bit store is single-copy atomic, so a
// reader that sees the sequence number it waits for has the payload that was stored with it -- no flag, no fence, no ordering
// assumption between stores (the LL protocol of the collective libraries).  A double travels as two words (low half, high half);
// a message of 4 doubles = 8 words = 64 bytes = one half of a 128-byte entry (the other half: the other iteration parity).
#ifndef EXA_P2P_SCOPE
#define EXA_P2P_SCOPE __HIP_MEMORY_SCOPE_SYSTEM
#endif
__device__ __forceinline__ void p2p_put(unsigned long long addr, double v, int seq)
{
    unsigned long long* const q = reinterpret_cast<unsigned long long*>(addr);
    const unsigned long long s = (unsigned long long)(unsigned)seq << 32;
    __hip_atomic_store(q, s | (unsigned)__double2loint(v), __ATOMIC_RELAXED, EXA_P2P_SCOPE);
    __hip_atomic_store(q + 1, s | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, EXA_P2P_SCOPE);
}
__device__ __forceinline__ bool p2p_get(const double* p, int seq, double& v)
{
    const unsigned long long* const q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, EXA_P2P_SCOPE), b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, EXA_P2P_SCOPE);
    v = __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
    return (unsigned)(a >> 32) == (unsigned)seq && (unsigned)(b >> 32) == (unsigned)seq;
}
// (what a failed wait leaves in rs.fail -- the host reports it: 1 a bus / decision of this GPU, 2 a cut branch's twin, 3 a peer's residual
//  sums, 4 a peer's end-of-launch status or a failure it reported, 5 a twin in the launch's last step, 8 a peer that never reached the launch)
// Waits INSIDE a resident launch are capped by a spin count (~ 1 s): every rank is known to be there (k_p2p_rendezvous), a word that does not come
// within a second will not come.  Waits of the HOST-ISSUED kernels -- the rendezvous in front of a launch, the sums exchanges between launches -- are
// PATIENT: a peer that is late is no failure (its host may be seconds behind ours: another call of the application between two of ours, a page fault,
// the first launch's code load), only one that never comes is; they are capped by the clock.  (Round 6: a sums exchange that gave up after 0.9 s
// handed its caller NaN -- and the outer loop, which compares against it, a penalty the other ranks did not take:
// profiles/r06_heavy_split_on_partitions.txt.)
constexpr long long P2P_PATIENT_TICKS = 60ll * 100000000ll;   // 60 s of the constant 100 MHz clock (wall_clock64)
// the 4 doubles of `entry` (parity seq & 1) carry `seq`: their values; false = the wait was capped / another wavefront has failed
__device__ __forceinline__ bool p2p_wait4(const double* entry, int seq, int* fail, double (&out)[4], int code = 2, long long patient_ticks = 0)
{
    const double* q = entry + 8 * (seq & 1);
    int spins = 0;
    long long t0 = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; k++) ok = p2p_get(q + 2 * k, seq, out[k]) && ok;
        if (ok) {
            asm volatile("" ::: "memory");
            return true;
        }
        ++spins;
        bool over = spins > (1 << 20);
        if (patient_ticks > 0) {
            if (spins == 1) t0 = wall_clock64();
            over = (spins & 255) == 0 && wall_clock64() - t0 > patient_ticks;
        }
        if (over || (fail != nullptr && (spins & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            if (fail != nullptr) {   // (the FIRST failure's code stays)
                int expected = 0;
                (void)__hip_atomic_compare_exchange_strong(fail, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return false;
        }
        __builtin_amdgcn_s_sleep(24);
    }
}
__device__ __forceinline__ void p2p_send4(unsigned long long entry_addr, int seq, const double (&v)[4], int delay_ticks = 0)
{
#if defined(EXA_TEST_HOOKS)
    if (delay_ticks > 0) {   // a slow link: the message leaves `delay_ticks` late (tests/test_gpu_dist_threads.py: what does an iteration lose?)
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)delay_ticks) __builtin_amdgcn_s_sleep(8);
    }
#else
    (void)delay_ticks;
#endif
    const unsigned long long q = entry_addr + 64ull * (unsigned)(seq & 1);
#pragma unroll
    for (int k = 0; k < 4; k++) p2p_put(q + 16ull * k, v[k], seq);
}

// One wavefront IN FRONT OF every resident launch of a partition with peers: the ranks wait for each other here, patiently (lane q talks to rank
// q through the status entries, the half the agreement kernel below does not use: sequence numbers ... ffe against ... fff), so that the launch that
// follows starts within microseconds of its peers' and its own short wait caps only ever meet real failures.  A rank that hears nothing for
// rp.patient_ticks marks the launch failed before it starts (code 8): its wavefronts then give up at their first wait, the agreement kernel tells
// whoever is there, every rank that is repeats the call with the two-launch loop.
__global__ void __launch_bounds__(64) k_p2p_rendezvous(ResidentPeers rp, int* fail)
{
    const int lane = threadIdx.x;
    const bool peer_lane = lane < rp.nranks && lane != rp.rank;
    const int seq = (rp.epoch << 12) | 0xffe;
    const double v[4] = {1.0, 0.0, 0.0, 0.0};
    double got[4] = {0.0, 0.0, 0.0, 0.0};
    bool bad = false;
    if (peer_lane) {
        p2p_send4(rp.sums_at[lane] + 128ull * (unsigned)rp.nranks, seq, v);
        bad = !p2p_wait4(rp.box + (size_t)P2P_ENTRY * (rp.ncut + rp.nranks + lane), seq, nullptr, got, 8, rp.patient_ticks);
    }
    if (__ballot(bad) != 0ull && lane == 0) {
        int expected = 0;
        (void)__hip_atomic_compare_exchange_strong(fail, &expected, 8, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One wavefront behind every resident launch of a partition with peers: the ranks tell each other how their launch ended (lane q talks
// to rank q: our rs.fail into its status entry, its into ours) and a rank that hears of a failure -- or hears nothing within the cap --
// marks its own launch failed: every rank then restores the iterate its call started from and repeats the call with the two-launch
// loop.  (A failure INSIDE the launch already spreads by itself: a rank that stops sending starves the twins of its cut lanes and
// the peers' control wavefronts, whose waits are capped.  This closes the window at the very end of a launch.)
__global__ void __launch_bounds__(64) k_p2p_agree(ResidentPeers rp, int* fail)
{
    const int lane = threadIdx.x;
    if (lane == 0) reinterpret_cast<unsigned long long*>(fail + 1)[1] = (unsigned long long)wall_clock64();   // diagnostics: the launch before us has ended
    const bool peer_lane = lane < rp.nranks && lane != rp.rank;
    const int seq = (rp.epoch << 12) | 0xfff;
    const int mine = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v[4] = {(double)mine, 0.0, 0.0, 0.0};
    double got[4] = {0.0, 0.0, 0.0, 0.0};
    bool bad = false;
    if (peer_lane) {
        p2p_send4(rp.sums_at[lane] + 128ull * (unsigned)rp.nranks, seq, v);
        // (no flag to watch: the launch's own must not cut this wait short)
        bad = !p2p_wait4(rp.box + (size_t)P2P_ENTRY * (rp.ncut + rp.nranks + lane), seq, nullptr, got) || got[0] != 0.0;
    }
    if (__ballot(bad) != 0ull && lane == 0 && mine == 0) __hip_atomic_store(fail, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The sums-only exchange of a partition whose peers' mailboxes are mapped (the z / l pass that completes a loop's last iteration; ||z|| of
// the outer loop): what k_halo_pack_sums + the grouped ncclSend / ncclRecv + k_halo_unpack do on the two-launch loop, in one wavefront
// through the mailboxes -- every rank's four sums to every rank, added IN RANK ORDER, then the loop control advanced or the totals
// delivered.  Host-issued exchanges have entries of their own (behind the status entries) and sequence numbers 0x80000000 | count,
// whose parity alternates the two halves of an entry.  sums_mode as k_halo_unpack: 0 add, 1 only if ctrl->pending says so.
// The wait is patient (rp.patient_ticks); one that runs into its cap all the same leaves *xfail = 1 -- host-visible memory: the host reads it at
// the call's synchronisation and fails the call (dist_exchange_failed: the peers are gone).
__global__ void __launch_bounds__(64) k_p2p_exchange4(ResidentPeers rp, unsigned xcount, const double* __restrict__ mine4, Ctrl* ctrl, double* out, int sums_mode, int* xfail)
{
    if (ctrl != nullptr && ctrl->done) return;
    if (sums_mode == 1 && ctrl != nullptr && !ctrl->pending) return;
    const int lane = threadIdx.x;
    const bool peer_lane = lane < rp.nranks && lane != rp.rank;
    const int seq = (int)(0x80000000u | xcount);
    double mine[4], got[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 4; j++) mine[j] = mine4[j];
    if (peer_lane) p2p_send4(rp.sums_at[lane] + 256ull * (unsigned)rp.nranks, seq, mine, rp.delay_ticks);
    bool ok = true;
    if (peer_lane) ok = p2p_wait4(rp.box + (size_t)P2P_ENTRY * (rp.ncut + 2 * rp.nranks + lane), seq, nullptr, got, 2, rp.patient_ticks);
    if (__ballot(!ok) != 0ull) {   // a peer is gone: the failure must be visible in what the caller reads -- the sums become NaN, the loop ends
        if (lane == 0) {
            *xfail = 1;
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            if (out != nullptr) out[0] = out[1] = out[2] = out[3] = nan;
            if (ctrl != nullptr) {
                ctrl->res[0] = ctrl->res[1] = ctrl->res[2] = ctrl->res[3] = nan;
                ctrl->done = 1;
            }
        }
        return;
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < rp.nranks; q++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double gq = __shfl(got[j], q, 64);
            s[j] += (q == rp.rank) ? mine[j] : gq;
        }
    }
    if (lane != 0) return;
    if (out != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; j++) out[j] = s[j];
    }
    if (ctrl != nullptr) ctrl_advance(ctrl, s);
}

// The results of an ADOPTED branch's last solve, from its heavy wavefront's hand-over into the planes of the buffer set that solve belongs to
// (dp: that set), by the branch's own lane (ResidentHeavy: adoption; u has been read already).  stamp: the resident iteration of that solve.
__device__ __forceinline__ void adopted_results_to_planes(const Dev& dp, const ResidentSync& rs, const double* ho, const double (&u)[8], int b, int base, int Lp, int stamp)
{
#pragma unroll
    for (int k = 0; k < 8; k++) dp.u[base + k * Lp] = u[k];
    dp.ls1[b] = ld_agent(ho + 8);
    dp.ls2[b] = ld_agent(ho + 9);
    dp.mu[b] = ld_agent(ho + 10);
    const long long pk = __double_as_longlong(ld_agent(ho + 11));
    const int ticks = (int)((pk >> 40) & 0xfffff);
    dp.st_auglag[b] = (int)(pk & 0xfffff);
    dp.st_minor[b] = (int)((pk >> 20) & 0xfffff);
    dp.st_ticks[b] = ticks;
    if (ticks >= 3) rs.last3[b] = (stamp << 4) | min(ticks, 15);
}

template <int TILES, bool HEAVY = false, bool PEERS = false>
__global__ void __launch_bounds__(BR_BLOCK, 1) k_resident_t(const Dev* __restrict__ dp0, const Dev* __restrict__ dp1, BranchOptions o, double scale,
                                                             Ctrl* ctrl, int gen_blocks, double baseMVA, int iters, ResidentSync rs, ResidentTiles rt,
                                                             ResidentHeavy rh, ResidentPeers rp)
{
    // (PEERS && HEAVY: heavy wavefronts take INTERIOR branches only -- k_heavy_select passes the cut branches by --, so a heavy wavefront
    //  never talks to a peer and its code is the same with and without peers)
    if (PEERS && blockIdx.x == 0 && threadIdx.x == 0)   // diagnostics: when this launch's first wavefront started (constant 100 MHz clock)
        reinterpret_cast<unsigned long long*>(rs.fail + 1)[0] = (unsigned long long)wall_clock64();
    if (ctrl->done) return;
    const int base_iter = ctrl->base;
    const double zl_beta = ctrl->beta;
    const bool first_has_zl = (base_iter + 1 != ctrl->first_iter);
    const int nheavy = (TILES == 1 && HEAVY) ? rh.n : 0;   // (HEAVY = false: the kernel without a line of the heavy wavefronts' code)
    const int nwaves = gridDim.x - 1 - nheavy;    // the last block is the control wavefront, the heavy wavefronts sit in front of it
    // contributions the control wavefront expects per z / l pass = tiles of the grid; wavefronts [0, nbr_waves) hold branches
    const int nwork = (TILES == 1) ? nwaves : rt.nb + rt.ng;
    const int nbr_waves = (TILES == 1) ? nwaves - gen_blocks : rt.wb;
    const int nbr_tiles = (TILES == 1) ? nbr_waves : rt.nb;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
        // ---------------- control wavefront: one decision per z / l pass
        const int npass = resident_passes(first_has_zl, iters - 1);
        for (int p = 1; p <= npass; p++) {
            if (!resident_wait(rs.contrib, p * nwork, rs.fail, 6)) return;
            const int i = p - (first_has_zl ? 1 : 0);     // index whose prologue made pass p
            const double* part = rs.partials[i & 1];
            double tot[4] = {0.0, 0.0, 0.0, 0.0};
            for (int vw = 0; vw < EW_BLOCK / 64; vw++) {   // the reduction block of the bus launch has EW_BLOCK threads: same order
                double sj[4] = {0.0, 0.0, 0.0, 0.0};
                for (int bq = vw * 64 + lane; bq < nwork; bq += EW_BLOCK) {
#pragma unroll
                    for (int j = 0; j < 4; j++) sj[j] += ld_agent(part + 4 * (size_t)bq + j);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) sj[j] += __shfl_down(sj[j], off, 64);
                    tot[j] += sj[j];   // (lane 0: 0 + wave 0 + wave 1 + ... as block_reduce4 adds them)
                }
            }
            if (PEERS) {
                // this rank's sums to every peer, the peers' sums from our mailbox (lane q talks to rank q), added in rank order
#pragma unroll
                for (int j = 0; j < 4; j++) tot[j] = __shfl(tot[j], 0, 64);
                double got[4] = {0.0, 0.0, 0.0, 0.0};
                const bool peer_lane = lane < rp.nranks && lane != rp.rank;
                const int seq = (rp.epoch << 12) | p;
                if (peer_lane) p2p_send4(rp.sums_at[lane], seq, tot, rp.delay_ticks);
                bool ok = true;
                if (peer_lane) ok = p2p_wait4(rp.box + (size_t)P2P_ENTRY * (rp.ncut + lane), seq, rs.fail, got, 3);
                if (__ballot(!ok) != 0ull) return;
                double sum[4] = {0.0, 0.0, 0.0, 0.0};
                for (int q = 0; q < rp.nranks; q++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const double gq = __shfl(got[j], q, 64);
                        sum[j] += (q == rp.rank) ? tot[j] : gq;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) tot[j] = sum[j];
            }
            int stop = 0;
            if (lane == 0) {
                ctrl_advance(ctrl, tot);
                stop = ctrl->done;
                __hip_atomic_store(rs.stop, stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (diagnostics; readers take the bit of the decision word)
                __hip_atomic_store(rs.decided, stop ? (p | RESIDENT_STOP_BIT) : p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stop = __shfl(stop, 0, 64);
            if (stop) return;
        }
        return;
    }
    __shared__ MuPowTable pw_lds;
    __shared__ double park_lds[(40 + EXA_PARK_SLOTS) * BR_BLOCK];
    if (TILES == 1 && HEAVY && (int)blockIdx.x >= nwaves) {
        // ---------------- heavy wavefront: one branch, 64 clones
        const int hw = (int)blockIdx.x - nwaves;
        int b = rh.list[hw], i_start = 0;
        if (b < 0) {
            // a FREE heavy wavefront: waits for a branch lane to hand it a straggler (ResidentHeavy: adoption), for the stop, or for the
            // launch's last decision (every request is out before the pass that decision is made from)
            const int npass_all = resident_passes(first_has_zl, iters - 1);
            if (rh.adopt_thr <= 0 || npass_all <= 0) return;
            int r = 0, spins = 0;
            for (;;) {
                r = __hip_atomic_load(rh.req + hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r != 0) break;
                if ((spins & 15) == 15) {
                    const int v = __hip_atomic_load(rs.decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v & RESIDENT_STOP_BIT) != 0 || __hip_atomic_load(rs.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (v >= npass_all) {
                        r = __hip_atomic_load(rh.req + hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (r == 0) return;
                        break;
                    }
                }
                ++spins;
                __builtin_amdgcn_s_sleep(64);
            }
            asm volatile("" ::: "memory");
            b = (r & 0xfffff) - 1;
            i_start = r >> 20;
        }
        const bool adopted = i_start > 0;
        if (lane < 24) reinterpret_cast<double*>(&pw_lds)[lane] = reinterpret_cast<const double*>(&c_mupow)[lane];
        double* const park_h = park_lds;
        double* const search_h = park_lds + 32 * BR_BLOCK;   // (entries 32 .. 39 hold z in the ordinary lanes' use of the area)
        double* const keep = search_h + 16;                  // what the wavefront needs again behind the solve, not in it: bnd, lz | z, l
        const MuPowTable& pw_h = pw_lds;
        const Dev& d0 = *dp0;
        const int Lp = d0.Lp, base = 2 * d0.Gp + b;
        const int4 bus4 = reinterpret_cast<const int4*>(d0.br_bus4)[b];
        const int rec_from = d0.rec_br[2 * b], rec_to = d0.rec_br[2 * b + 1];
        BranchData bd;
        double zc[8], lc[8], uc[8];
        double x[6], phi[4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int e = base + k * Lp;
            bd.rho_[k] = d0.rho[e];
            bd.Y_[k] = d0.Y[b + k * Lp];
            if (lane == 0) {
                keep[k] = d0.bnd[b + k * Lp];
                keep[8 + k] = d0.lz[e];
            }
            if (!adopted) {
                zc[k] = rh.init[16 * hw + k];
                lc[k] = rh.init[16 * hw + 8 + k];
                uc[k] = (d0.u_in ? d0.u_in : d0.u)[e];
            } else {   // (what the branch's lane left in its request: the state behind index i_start - 1)
                const double* const rec = rh.adopt + 32 * (size_t)hw;
                uc[k] = ld_agent(rec + k);
                zc[k] = ld_agent(rec + 8 + k);
                lc[k] = ld_agent(rec + 16 + k);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!adopted) {
            bd.ls1 = (d0.ls1_in ? d0.ls1_in : d0.ls1)[b];
            bd.ls2 = (d0.ls2_in ? d0.ls2_in : d0.ls2)[b];
            bd.mu = (d0.mu_in ? d0.mu_in : d0.mu)[b];
        } else {
            const double* const rec = rh.adopt + 32 * (size_t)hw;
            bd.ls1 = ld_agent(rec + 24);
            bd.ls2 = ld_agent(rec + 25);
            bd.mu = ld_agent(rec + 26);
        }
        bd.scale = scale;
        const double rate_0 = d0.rateA[b];
        // The bus updates of the last iteration that THIS wavefront completed (its lanes 0 and 1 arrive at the two buses side by side;
        // the launch's slowest branch is the last to arrive at both of its buses): kept in registers -- the same doubles it published --,
        // no wait for a flag it raised itself, no read-back.  (wave-uniform)
        double keep_f[4] = {0.0, 0.0, 0.0, 0.0}, keep_t[4] = {0.0, 0.0, 0.0, 0.0};
        bool have_f = false, have_t = false;
        for (int i = i_start; i < iters; i++) {
#if defined(EXA_PROFILE_PHASES)
            unsigned long long hts[8], hx[4] = {0ull, 0ull, 0ull, 0ull};   // diagnostic build: stamps of this iteration of the heavy wavefront (scripts/heavy_timeline.py)
            hts[0] = __builtin_amdgcn_s_memtime();
            hts[6] = (unsigned long long)((have_f ? 1 : 0) | (have_t ? 2 : 0));
#endif
            // The heavy wavefront does NOT wait for the stop decision here.  What the decision protects is the iterate that would be final:
            // the ordinary lanes' z / l pass overwrites z, l in place, and an x-update overwrites the u of the buffer set two indices back.
            // A heavy wavefront keeps z, l in registers and stores nothing before its solve is over: it reads its two bus results, solves
            // on speculation and asks for the decision in front of its stores (below) -- by then it has been out for a whole solve, where
            // at the top it cost the launch's critical path the full latency of arrival -> bus -> tile's z / l pass -> contribution ->
            // control wavefront -> decision in every iteration (profiles/r05_heavy_wavefront_series.txt: 15 - 27 k of 115 k clocks).
            if (i > 0 && !resident_top_wait(rs, 0, have_f ? nullptr : rs.ready + bus4.x, have_t ? nullptr : rs.ready + bus4.z, i)) return;   // (wave-uniform)
#if defined(EXA_PROFILE_PHASES)
            hts[1] = __builtin_amdgcn_s_memtime();   // decision + flags seen
#endif
            const Dev& d = (i & 1) ? *dp1 : *dp0;
            const bool do_zl = (i > 0) || first_has_zl;
            double vv[8];
            if (i > 0) {
                double rf[4], rt4[4];
                if (have_f && have_t) {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        rf[m] = keep_f[m];
                        rt4[m] = keep_t[m];
                    }
                } else {   // (a result this wavefront published itself reads back as what it kept: its stores were acknowledged before its flag went up)
                    const double* pf = rs.busres[i & 1] + 4 * (size_t)bus4.x;
                    const double* pt = rs.busres[i & 1] + 4 * (size_t)bus4.z;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        rf[m] = ld_agent(pf + m);
                        rt4[m] = ld_agent(pt + m);
                    }
                }
                vv[0] = (uc[0] + zc[0]) + (lc[0] + rf[0]) / bd.rho_[0];
                vv[1] = (uc[1] + zc[1]) + (lc[1] + rf[1]) / bd.rho_[1];
                vv[4] = rf[2];
                vv[6] = rf[3];
                vv[2] = (uc[2] + zc[2]) + (lc[2] + rt4[0]) / bd.rho_[2];
                vv[3] = (uc[3] + zc[3]) + (lc[3] + rt4[1]) / bd.rho_[3];
                vv[5] = rt4[2];
                vv[7] = rt4[3];
            } else {
                const double* const vbuf = d.v_in ? d.v_in : d.v;
#pragma unroll
                for (int k = 0; k < 8; k++) vv[k] = vbuf[base + k * Lp];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (do_zl) {   // (the values the tile's lane forms, stores and adds to the residual sums)
                    double zs[4] = {0.0, 0.0, 0.0, 0.0}, zn;
                    zl_values(uc[k], vv[k], bd.rho_[k], zl_beta, keep[8 + k], zc[k], lc[k], zs, bd.lam_[k], zn);
                    bd.xt_[k] = vv[k] - zn;
                    zc[k] = zn;
                    lc[k] = bd.lam_[k];
                } else {
                    bd.lam_[k] = lc[k];
                    bd.xt_[k] = vv[k] - zc[k];
                }
            }
            if (base_iter + i + 1 == 1) bd.mu = 10.0;
            SymMat<6> A, L;
            double xl[6], xu[6];
            {
                double bnd[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    bnd[k] = keep[k];
                    if (lane == 0) {
                        keep[16 + k] = zc[k];
                        keep[24 + k] = lc[k];
                    }
                }
                branch_start(uc, bnd, rate_0, x, xl, xu);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(EXA_PROFILE_PHASES)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hts[2]) : "v"(x[0]), "v"(bd.xt_[7]) : "memory");   // bus results in, v, z / l, start point formed
#endif
            BranchStats st;
            st.tl = nullptr;
            branch_solve<true>(bd, o, pw_h, x, xl, xu, A, L, phi, st, park_h + lane, search_h, EXA_PARKER(park_lds + 40 * BR_BLOCK + lane));
            asm volatile("" ::: "memory");
#if defined(EXA_PROFILE_PHASES)
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hts[3]) : "v"(phi[0]) : "memory");   // solve over
#endif
            // the decision that the ordinary lanes read at the top of this index: if the loop has stopped, this index never ran -- nothing of it
            // has left the wavefront yet (the u of buffer set i & 1, which these stores would overwrite, belongs to the final iterate)
            if (i > 0 && !resident_go_on(rs, resident_passes(first_has_zl, i - 1))) return;   // (wave-uniform)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                zc[k] = keep[16 + k];
                lc[k] = keep[24 + k];
            }
            uc[0] = phi[0];
            uc[1] = phi[1];
            uc[2] = phi[2];
            uc[3] = phi[3];
            uc[4] = x[0] * x[0];
            uc[5] = x[1] * x[1];
            uc[6] = x[2];
            uc[7] = x[3];
            if (lane == 0) {
                double* const ho = rh.u + ((size_t)(i & 1) * nheavy + hw) * 16;   // (read by the tile's lane with its next z / l pass)
#pragma unroll
                for (int k = 0; k < 8; k++) st_agent(ho + k, uc[k]);
                if (adopted) {
                    // an ADOPTED branch: its lane wrote these planes' words earlier in this launch, so the lane goes on writing them -- from
                    // this hand-over, with its next z / l pass (one word dirty in two XCDs' L2s leaves the result to the order of the
                    // write-backs: seen with the first form of the adoption, which let the heavy wavefront write them)
                    st_agent(ho + 8, bd.ls1);
                    st_agent(ho + 9, bd.ls2);
                    st_agent(ho + 10, bd.mu);
                    st_agent(ho + 11, __longlong_as_double((long long)(unsigned)st.auglag_it | ((long long)(unsigned)st.minor_it << 20) | ((long long)(unsigned)st.ticks << 40)));
                } else {
                    // (a branch chosen in front of the launch: the planes' words of this branch are written here only, the tile's lane leaves them
                    //  alone.  They share cache lines with the neighbouring branches' words, which the tile's wavefront writes from another XCD --
                    //  different bytes of a line dirty in two L2s, merged by the byte masks of the write-backs like the output of any two workgroups)
#pragma unroll
                    for (int k = 0; k < 8; k++) d.u[base + k * Lp] = uc[k];
                    d.ls1[b] = bd.ls1;
                    d.ls2[b] = bd.ls2;
                    d.mu[b] = bd.mu;
                    d.st_auglag[b] = st.auglag_it;
                    d.st_minor[b] = st.minor_it;
                    d.st_ticks[b] = st.ticks;
                    if (st.ticks >= 3) rs.last3[b] = ((rs.stamp0 + i + 1) << 4) | min(st.ticks, 15);
                }
#if defined(EXA_PROFILE_PHASES)
                for (int q = 0; q < 13; q++) d.st_cyc[q * d.Lp + b] = st.cyc[q] + (q == 12 ? 1000ull : 0ull);   // (phase clocks of the heavy wavefront's last solve; entry 12 = 1000 + trips marks it)
                asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hx[0]) : : "memory");   // hand-over, planes, diagnostics acknowledged (the shipped build does not wait here)
#endif
                double* const tm = rs.terms[(i + 1) & 1];
                {
                    double* q = tm + 4 * (size_t)rec_from;
                    st_agent(q, (uc[0] + zc[0]) + (bd.lam_[0] / bd.rho_[0]));
                    st_agent(q + 1, (uc[1] + zc[1]) + (bd.lam_[1] / bd.rho_[1]));
                    st_agent(q + 2, bd.lam_[4] + bd.rho_[4] * (uc[4] + zc[4]));
                    st_agent(q + 3, bd.lam_[6] + bd.rho_[6] * (uc[6] + zc[6]));
                }
                {
                    double* q = tm + 4 * (size_t)rec_to;
                    st_agent(q, (uc[2] + zc[2]) + (bd.lam_[2] / bd.rho_[2]));
                    st_agent(q + 1, (uc[3] + zc[3]) + (bd.lam_[3] / bd.rho_[3]));
                    st_agent(q + 2, bd.lam_[5] + bd.rho_[5] * (uc[5] + zc[5]));
                    st_agent(q + 3, bd.lam_[7] + bd.rho_[7] * (uc[7] + zc[7]));
                }
            }
            resident_order_stores();
#if defined(EXA_PROFILE_PHASES)
            hts[4] = __builtin_amdgcn_s_memtime();   // results, hand-over and terms stored (acknowledged)
#endif
            {
                // lane 0 arrives at the from-bus, lane 1 at the to-bus -- side by side: one counter round trip, one round of term loads, one
                // chain, one acknowledgement for both (one after the other on lane 0 they were 2 x 2.2 us of every iteration)
                double res[4] = {0.0, 0.0, 0.0, 0.0};
                bool last = false;
#if defined(EXA_PROFILE_PHASES)
                if (lane < 2) last = resident_arrive_keep(d, rs, lane == 0 ? bus4.x : bus4.z, lane == 0 ? bus4.y : bus4.w, i + 1, res, lane == 0 ? hx + 1 : nullptr);
#else
                if (lane < 2) last = resident_arrive_keep(d, rs, lane == 0 ? bus4.x : bus4.z, lane == 0 ? bus4.y : bus4.w, i + 1, res);
#endif
                const unsigned long long lm = __ballot(last);
                have_f = (lm & 1ull) != 0ull;
                have_t = (lm & 2ull) != 0ull;
#pragma unroll
                for (int m = 0; m < 4; m++) {
#if defined(__HIP_DEVICE_COMPILE__)
                    keep_f[m] = lane_value(res[m], 0);
                    keep_t[m] = lane_value(res[m], 1);
#else
                    keep_f[m] = keep_t[m] = res[m];   // (host pass of the compiler: never run)
#endif
                }
            }
#if defined(EXA_PROFILE_PHASES)
            hts[5] = __builtin_amdgcn_s_memtime();   // arrivals (and the bus updates the wavefront completed) done
#if defined(EXA_HEAVY_SERIES)
            if (d.st_tl && lane == 0 && hw < nwaves - gen_blocks && i >= iters - 16) {   // series build: the last 16 iterations of the launch
                unsigned long long* q = d.st_tl + (size_t)hw * 128 + 32 + 3 * (i - (iters - 16));
                const unsigned long long m20 = (1ull << 20) - 1;
                q[0] = hts[0];
                q[1] = ((hts[3] - hts[2]) << 8) | (unsigned long long)st.ticks;
                q[2] = ((hts[1] - hts[0]) & m20) | (((hts[4] - hts[3]) & m20) << 20) | (((hts[5] - hts[4]) & m20) << 40) | (hts[6] << 60);   // top wait | stores acked | arrivals | kept bits
            }
            if (false) {
#else
            if (d.st_tl && lane == 0 && nwaves - gen_blocks > 0) {   // the last two iterations survive, in the unused part of a tile's timeline block
#endif
                const int nbt = nwaves - gen_blocks;
                unsigned long long* q = d.st_tl + (size_t)(hw % nbt) * 128 + 32 + 16 * ((hw / nbt) & 3) + 8 * (i & 1);
                for (int k = 0; k < 6; k++) q[k] = hts[k];
                q[6] = hts[6] | ((unsigned long long)st.ticks << 8);
                q[7] = (unsigned long long)i + 1;
                if (hw < nbt) {
                    unsigned long long* q2 = d.st_tl + (size_t)hw * 128 + 96 + 8 * (i & 1);
                    for (int k = 0; k < 4; k++) q2[k] = hx[k];
                }
            }
#endif
        }
        // (v of the last index is written by the tile's lane, which wrote that word before: two XCDs with the same bytes dirty in their
        //  L2s would leave the order of the write-backs to chance -- seen)
        return;
    }
    if ((int)blockIdx.x >= nbr_waves) {
        // ---------------- generator lanes
        const int gw = (int)blockIdx.x - nbr_waves;
        int i = 0;
        for (; i < iters; i++) {
            if (i > 0 && !resident_go_on(rs, resident_passes(first_has_zl, i - 1))) return;   // (wave-uniform)
            const Dev& d = (i & 1) ? *dp1 : *dp0;
            const bool do_zl = (i > 0) || first_has_zl;
            for (int t = 0; t < TILES; t++) {
                const int gt = gw + t * ((TILES == 1) ? 0 : rt.wg);
                if (TILES > 1 && gt >= rt.ng) break;
                const int gi = gt * BR_BLOCK + lane;
                const bool on = gi < d.G;
                const int g = on ? gi : 0;
                const int I = d.gen_bus2[2 * g], cnt = d.gen_bus2[2 * g + 1];
                double zs[4] = {0.0, 0.0, 0.0, 0.0};
                if (on) {
                    const double* const uin = d.u_in ? d.u_in : d.u;
                    double* const vbuf = const_cast<double*>(d.v_in ? d.v_in : d.v);
                    const int pg = g, qg = d.Gp + g;
                    const double rho_p = d.rho[pg], rho_q = d.rho[qg];
                    double lp = d.l[pg], lq = d.l[qg], zp = d.z[pg], zq = d.z[qg];
                    if (do_zl) {
                        const double up0 = uin[pg], uq0 = uin[qg];
                        double vp, vq;
                        if (i > 0) {
                            if (!resident_wait(rs.ready + I, i, rs.fail)) return;
                            const double* r = rs.busres[i & 1] + 4 * (size_t)I;
                            vp = (up0 + zp) + (lp - ld_agent(r)) / rho_p;
                            vq = (uq0 + zq) + (lq - ld_agent(r + 1)) / rho_q;
                            vbuf[pg] = vp;
                            vbuf[qg] = vq;
                        } else {   // first index of a later batch: the previous launch left v in memory
                            vp = vbuf[pg];
                            vq = vbuf[qg];
                        }
                        zl_element(d, pg, up0, vp, rho_p, zl_beta, zs, lp, zp);
                        zl_element(d, qg, uq0, vq, rho_q, zl_beta, zs, lq, zq);
                    }
                    double up, uq;
                    generator_update(d, baseMVA, g, vbuf, up, uq);
                    double* q = rs.terms[(i + 1) & 1] + 4 * (size_t)d.rec_gen[g];
                    st_agent(q, (up + zp) + (lp / rho_p));
                    st_agent(q + 1, (uq + zq) + (lq / rho_q));
                }
                if (do_zl) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) zs[j] += __shfl_down(zs[j], off, 64);
                    }
                    if (lane == 0) resident_contribute(rs, i & 1, nbr_tiles + gt, zs);
                }
                resident_order_stores();
                if (on) resident_arrive(d, rs, I, cnt, i + 1);
            }
        }
        for (int t = 0; t < TILES; t++) {   // v of the last index (what its bus launch would have written)
            const Dev& d = ((iters - 1) & 1) ? *dp1 : *dp0;
            const int gt = gw + t * ((TILES == 1) ? 0 : rt.wg);
            if (TILES > 1 && gt >= rt.ng) break;
            const int gi = gt * BR_BLOCK + lane;
            if (gi < d.G) {
                const int I = d.gen_bus2[2 * gi];
                if (!resident_wait(rs.ready + I, iters, rs.fail)) return;
                const double* r = rs.busres[iters & 1] + 4 * (size_t)I;
                const int pg = gi, qg = d.Gp + gi;
                d.v[pg] = (d.u[pg] + d.z[pg]) + (d.l[pg] - ld_agent(r)) / d.rho[pg];
                d.v[qg] = (d.u[qg] + d.z[qg]) + (d.l[qg] - ld_agent(r + 1)) / d.rho[qg];
            }
        }
        return;
    }
    // ---------------- branch lanes
    if (lane < 24) reinterpret_cast<double*>(&pw_lds)[lane] = reinterpret_cast<const double*>(&c_mupow)[lane];
    const int Lp = dp0->Lp;
    const int my_tile0 = (TILES == 1) ? (int)blockIdx.x : rt.tiles[2 * blockIdx.x];
    const int my_tile1 = (TILES == 1) ? -1 : rt.tiles[2 * blockIdx.x + 1];
    for (int i = 0; i < iters; i++) {
#if defined(EXA_PROFILE_PHASES)
        unsigned long long ts[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};   // diagnostic build: stamps of this iteration (scripts/resident_timeline.py)
        ts[0] = __builtin_amdgcn_s_memtime();
#endif
        if (i > 0 && !resident_go_on(rs, resident_passes(first_has_zl, i - 1))) return;   // (wave-uniform: the loop has stopped)
        const Dev& d = (i & 1) ? *dp1 : *dp0;
        for (int t = 0; t < TILES; t++) {
            const int tile = (t == 0) ? my_tile0 : my_tile1;
            if (TILES > 1 && tile < 0) break;
            const int slot = tile * BR_BLOCK + lane;
            const bool lane_on = slot < d.L;
            const int b = lane_on ? slot : 0;
            const int base = 2 * d.Gp + b;
            const int4 bus4 = reinterpret_cast<const int4*>(d.br_bus4)[b];
            // (TILES = 1) the lane's branch is solved by a heavy wavefront in this launch: z / l pass, residual terms and v here, nothing
            // else.  No value of its own across the solve: rec_from = -1 says so, rec_to holds the heavy wavefront's number.
            const int hflag_raw = (TILES == 1 && HEAVY && nheavy > 0) ? rh.flag[b] : 0;
            const int hflag = hflag_raw & 0xfffff;
            const bool heavy_lane = lane_on && hflag != 0;
            // (adopted at the end of the last index: the u of that index is in the planes -- this lane wrote it --, not in the hand-over)
            const bool heavy_fresh = heavy_lane && ((hflag_raw >> 20) & 1) != 0;
            const bool heavy_mine = heavy_lane && ((hflag_raw >> 21) & 1) != 0;   // adopted in this launch: the planes' words of the branch stay this lane's
            if (TILES == 1 && HEAVY && heavy_fresh) rh.flag[b] = hflag | (1 << 21);
            const int rec_from = heavy_lane ? -1 : d.rec_br[2 * b], rec_to = heavy_lane ? hflag - 1 : d.rec_br[2 * b + 1];
            const int major_iter = base_iter + i + 1;
            const bool do_zl = (i > 0) || first_has_zl;
            const bool v_from_bus = (i > 0);
            const double* const uin = d.u_in ? d.u_in : d.u;
            double* const vbuf = const_cast<double*>(d.v_in ? d.v_in : d.v);
            double zs[4] = {0.0, 0.0, 0.0, 0.0};
            double ub[8], bnd[8], vv[8], zo[8], lo[8], lzv[8];
            BranchData bd;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = base + k * Lp;
                ub[k] = uin[e];
                bd.rho_[k] = d.rho[e];
                zo[k] = d.z[e];
                lo[k] = d.l[e];
                lzv[k] = do_zl ? d.lz[e] : 0.0;
                vv[k] = v_from_bus ? 0.0 : vbuf[e];
                bd.Y_[k] = d.Y[b + k * Lp];
                bnd[k] = d.bnd[b + k * Lp];
            }
            bd.ls1 = (d.ls1_in ? d.ls1_in : d.ls1)[b];
            bd.ls2 = (d.ls2_in ? d.ls2_in : d.ls2)[b];
            const double mu_0 = (d.mu_in ? d.mu_in : d.mu)[b];
            const double rate_0 = d.rateA[b];
#if defined(EXA_PROFILE_PHASES)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[1]) : "v"(rate_0) : "memory");   // loads landed
#endif
            // (PEERS) a cut branch: the bus of one of its ends is the peer's -- its update arrives through the mailbox, from the twin lane
            const int ccode = PEERS ? rp.cut_code[b] : -1;
            const bool rem_f = PEERS && ccode >= 0 && (ccode & 1) == 0, rem_t = PEERS && ccode >= 0 && (ccode & 1) == 1;
            if (PEERS && v_from_bus && lane_on) {
                double rf[4] = {0.0, 0.0, 0.0, 0.0}, rt4[4] = {0.0, 0.0, 0.0, 0.0};
                if (!rem_f) {
                    if (!resident_wait(rs.ready + bus4.x, i, rs.fail)) return;
                    const double* pf = rs.busres[i & 1] + 4 * (size_t)bus4.x;
#pragma unroll
                    for (int m = 0; m < 4; m++) rf[m] = ld_agent(pf + m);
                }
                if (!rem_t) {
                    if (!resident_wait(rs.ready + bus4.z, i, rs.fail)) return;
                    const double* pt = rs.busres[i & 1] + 4 * (size_t)bus4.z;
#pragma unroll
                    for (int m = 0; m < 4; m++) rt4[m] = ld_agent(pt + m);
                }
                if (TILES == 1 && HEAVY && heavy_lane && !heavy_fresh) {   // (as without peers: u of the last iteration from the branch's heavy wavefront)
                    const double* const ho = rh.u + ((size_t)((i - 1) & 1) * nheavy + rec_to) * 16;
#pragma unroll
                    for (int k = 0; k < 8; k++) ub[k] = ld_agent(ho + k);
                    if (heavy_mine) adopted_results_to_planes((i & 1) ? *dp0 : *dp1, rs, ho, ub, b, base, Lp, rs.stamp0 + i);
                }
                // Our bus's update to the twin lane FIRST -- every cut lane of the wavefront, in one region that only stores -- and only then
                // the wait for the twin's.  (With the send and the wait of the two kinds of cut lanes in the two arms of one branch, a
                // wavefront runs one arm to its end -- wait included -- before the other arm has sent: two wavefronts on two ranks that
                // each hold both kinds then wait for each other's second arm.  Met on the first run.)
                {
                    const int seq = (rp.epoch << 12) | i;
                    double out4[4], in4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int m = 0; m < 4; m++) out4[m] = rem_f ? rt4[m] : rf[m];
                    if (ccode >= 0) p2p_send4(rp.twin[ccode >> 1], seq, out4, rp.delay_ticks);
                    bool ok = true;
                    if (ccode >= 0) ok = p2p_wait4(rp.box + (size_t)P2P_ENTRY * (ccode >> 1), seq, rs.fail, in4);
                    if (!ok) return;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        rf[m] = rem_f ? in4[m] : rf[m];
                        rt4[m] = rem_t ? in4[m] : rt4[m];
                    }
                }
                vv[0] = (ub[0] + zo[0]) + (lo[0] + rf[0]) / bd.rho_[0];
                vv[1] = (ub[1] + zo[1]) + (lo[1] + rf[1]) / bd.rho_[1];
                vv[4] = rf[2];
                vv[6] = rf[3];
                vv[2] = (ub[2] + zo[2]) + (lo[2] + rt4[0]) / bd.rho_[2];
                vv[3] = (ub[3] + zo[3]) + (lo[3] + rt4[1]) / bd.rho_[3];
                vv[5] = rt4[2];
                vv[7] = rt4[3];
#pragma unroll
                for (int k = 0; k < 8; k++) vbuf[base + k * Lp] = vv[k];
            }
            if (!PEERS && v_from_bus && lane_on) {
                if (!resident_wait2(rs.ready + bus4.x, rs.ready + bus4.z, i, rs.fail)) return;
#if defined(EXA_PROFILE_PHASES)
                ts[2] = __builtin_amdgcn_s_memtime();   // both buses ready
#endif
                if (TILES == 1 && HEAVY && heavy_lane && !heavy_fresh) {
                    // the results of the last iteration were handed over by the branch's heavy wavefront before it arrived: u for this
                    // z / l pass, and everything into the planes of the set that iteration wrote
                    const double* const ho = rh.u + ((size_t)((i - 1) & 1) * nheavy + rec_to) * 16;
#pragma unroll
                    for (int k = 0; k < 8; k++) ub[k] = ld_agent(ho + k);
                    if (heavy_mine) adopted_results_to_planes((i & 1) ? *dp0 : *dp1, rs, ho, ub, b, base, Lp, rs.stamp0 + i);
                }
                const double* pf = rs.busres[i & 1] + 4 * (size_t)bus4.x;
                const double* pt = rs.busres[i & 1] + 4 * (size_t)bus4.z;
                const double rf[4] = {ld_agent(pf), ld_agent(pf + 1), ld_agent(pf + 2), ld_agent(pf + 3)};
                const double rt4[4] = {ld_agent(pt), ld_agent(pt + 1), ld_agent(pt + 2), ld_agent(pt + 3)};
                vv[0] = (ub[0] + zo[0]) + (lo[0] + rf[0]) / bd.rho_[0];
                vv[1] = (ub[1] + zo[1]) + (lo[1] + rf[1]) / bd.rho_[1];
                vv[4] = rf[2];
                vv[6] = rf[3];
                vv[2] = (ub[2] + zo[2]) + (lo[2] + rt4[0]) / bd.rho_[2];
                vv[3] = (ub[3] + zo[3]) + (lo[3] + rt4[1]) / bd.rho_[3];
                vv[5] = rt4[2];
                vv[7] = rt4[3];
#pragma unroll
                for (int k = 0; k < 8; k++) vbuf[base + k * Lp] = vv[k];   // (what the bus launch of the previous index would have written)
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (do_zl && lane_on) {
                    const int e = base + k * Lp;
                    double zn;
                    zl_values(ub[k], vv[k], bd.rho_[k], zl_beta, lzv[k], zo[k], lo[k], zs, bd.lam_[k], zn);
                    d.zp[e] = zo[k];
                    d.z[e] = zn;
                    d.l[e] = bd.lam_[k];
                    bd.xt_[k] = vv[k] - zn;
                    zo[k] = zn;
                } else {
                    bd.lam_[k] = lo[k];
                    bd.xt_[k] = vv[k] - zo[k];
                }
                park_lds[(32 + k) * BR_BLOCK + lane] = zo[k];
            }
            if (do_zl) {
                const bool counts = lane_on && (b < d.Lown);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!counts) zs[j] = 0.0;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) zs[j] += __shfl_down(zs[j], off, 64);
                }
                if (lane == 0) resident_contribute(rs, i & 1, tile, zs);
            }
#if defined(EXA_PROFILE_PHASES)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[3]) : : "memory");   // v, z / l pass, contribution done
#endif
            int adopt_w = -1;   // (>= 0: this lane has just handed its branch to free heavy wavefront adopt_w: ResidentHeavy, adoption)
            if (lane_on && !(TILES == 1 && HEAVY && heavy_lane)) {
                bd.mu = (major_iter == 1) ? 10.0 : mu_0;
                bd.scale = scale;
                SymMat<6> A, L;
                double x[6], xl[6], xu[6], phi[4];
                branch_start(ub, bnd, rate_0, x, xl, xu);
                BranchStats st;
                st.tl = nullptr;
                branch_solve(bd, o, pw_lds, x, xl, xu, A, L, phi, st, park_lds + lane, nullptr, EXA_PARKER(park_lds + 40 * BR_BLOCK + lane));
#if defined(EXA_PROFILE_PHASES)
                ts[4] = __builtin_amdgcn_s_memtime();   // (of the lanes that reach this point last: the wavefront's solve is over)
#endif
                d.u[base] = phi[0];
                d.u[base + Lp] = phi[1];
                d.u[base + 2 * Lp] = phi[2];
                d.u[base + 3 * Lp] = phi[3];
                d.u[base + 4 * Lp] = x[0] * x[0];
                d.u[base + 5 * Lp] = x[1] * x[1];
                d.u[base + 6 * Lp] = x[2];
                d.u[base + 7 * Lp] = x[3];
                double zz[8];
#pragma unroll
                for (int k = 0; k < 8; k++) zz[k] = park_lds[(32 + k) * BR_BLOCK + lane];
                const double un[8] = {phi[0], phi[1], phi[2], phi[3], x[0] * x[0], x[1] * x[1], x[2], x[3]};
                double* const tm = rs.terms[(i + 1) & 1];
                if (!PEERS || !rem_f) {   // (the end on a peer's bus has no slot here: the peer's replica of the branch pushes it there)
                    double* q = tm + 4 * (size_t)rec_from;
                    st_agent(q, (un[0] + zz[0]) + (bd.lam_[0] / bd.rho_[0]));
                    st_agent(q + 1, (un[1] + zz[1]) + (bd.lam_[1] / bd.rho_[1]));
                    st_agent(q + 2, bd.lam_[4] + bd.rho_[4] * (un[4] + zz[4]));
                    st_agent(q + 3, bd.lam_[6] + bd.rho_[6] * (un[6] + zz[6]));
                }
                if (!PEERS || !rem_t) {
                    double* q = tm + 4 * (size_t)rec_to;
                    st_agent(q, (un[2] + zz[2]) + (bd.lam_[2] / bd.rho_[2]));
                    st_agent(q + 1, (un[3] + zz[3]) + (bd.lam_[3] / bd.rho_[3]));
                    st_agent(q + 2, bd.lam_[5] + bd.rho_[5] * (un[5] + zz[5]));
                    st_agent(q + 3, bd.lam_[7] + bd.rho_[7] * (un[7] + zz[7]));
                }
                d.ls1[b] = bd.ls1;
                d.ls2[b] = bd.ls2;
                d.mu[b] = bd.mu;
                d.st_auglag[b] = st.auglag_it;
                d.st_minor[b] = st.minor_it;
                d.st_ticks[b] = st.ticks;
                if (TILES == 1 && st.ticks >= 3) rs.last3[b] = ((rs.stamp0 + i + 1) << 4) | min(st.ticks, 15);
                if (TILES == 1 && HEAVY && rh.adopt_thr > 0 && st.ticks >= rh.adopt_thr && i + 1 < iters && (!PEERS || ccode < 0)) {
                    // a straggler: from the next index on one of the launch's free heavy wavefronts solves this branch (ResidentHeavy: adoption)
                    const int w = rh.cnt[0] + __hip_atomic_fetch_add(rh.cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (w < nheavy) {
                        double* const rec = rh.adopt + 32 * (size_t)w;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            st_agent(rec + k, un[k]);
                            st_agent(rec + 8 + k, zz[k]);
                            st_agent(rec + 16 + k, bd.lam_[k]);
                        }
                        st_agent(rec + 24, bd.ls1);
                        st_agent(rec + 25, bd.ls2);
                        st_agent(rec + 26, bd.mu);
                        rh.flag[b] = (w + 1) | (1 << 20) | (1 << 21);
                        adopt_w = w;
                    }
                }
#if defined(EXA_PROFILE_PHASES)
                for (int q = 0; q < 13; q++) d.st_cyc[q * d.Lp + b] = st.cyc[q];   // (phase clocks of the lane's last solve)
#endif
            }
            resident_order_stores();
            if (TILES == 1 && HEAVY && adopt_w >= 0)   // (behind the acknowledgement of the record: the request itself)
                __hip_atomic_store(rh.req + adopt_w, (b + 1) | ((i + 1) << 20), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(EXA_PROFILE_PHASES)
            ts[5] = __builtin_amdgcn_s_memtime();   // results and terms stored
#endif
            if (PEERS) {
                const bool solved_here = lane_on && !(TILES == 1 && HEAVY && rec_from < 0 && ccode < 0);   // (a heavy wavefront's branch arrives from there)
                if (solved_here && !rem_f) resident_arrive(d, rs, bus4.x, bus4.y, i + 1);
                if (solved_here && !rem_t) resident_arrive(d, rs, bus4.z, bus4.w, i + 1);
            } else if (lane_on && !(TILES == 1 && HEAVY && rec_from < 0)) {
                resident_arrive(d, rs, bus4.x, bus4.y, i + 1);
                resident_arrive(d, rs, bus4.z, bus4.w, i + 1);
            }
#if defined(EXA_PROFILE_PHASES)
            ts[6] = __builtin_amdgcn_s_memtime();   // arrivals (and the bus updates this wavefront's lanes completed) done
#if defined(EXA_HEAVY_SERIES)
            if (TILES == 1 && d.st_tl && i >= iters - 16) {   // series build: the last 16 iterations of every tile (scripts/heavy_series.py)
                const bool solved = lane_on && !(TILES == 1 && HEAVY && heavy_lane);
                int mt = solved ? d.st_ticks[b] : 0;
                for (int off = 32; off > 0; off >>= 1) mt = max(mt, __shfl_xor(mt, off, 64));
                const unsigned long long sm = __ballot(solved);
                const int wl = sm ? (int)__ffsll((long long)sm) - 1 : 0;
                if (lane == wl) {
                    const unsigned long long m20 = (1ull << 20) - 1;
                    unsigned long long* q = d.st_tl + (size_t)blockIdx.x * 128 + 80 + 3 * (i - (iters - 16));
                    q[0] = ts[0];
                    q[1] = ((ts[2] - ts[0]) & m20) | (((ts[3] - ts[2]) & m20) << 20) | (((ts[4] - ts[3]) & m20) << 40);   // top -> buses ready | -> contributed | solve
                    q[2] = ((ts[6] - ts[4]) & m20) | ((unsigned long long)mt << 20) | ((unsigned long long)__popcll(sm) << 32);   // stores + arrivals | max trips | lanes that solved
                }
            }
#endif
            if (TILES == 1 && d.st_tl && lane == 0) {   // the last two iterations of the launch survive
                unsigned long long* q = d.st_tl + (size_t)blockIdx.x * 128 + 8 * (i & 1);
                for (int k = 0; k < 7; k++) q[k] = ts[k];
                q[7] = (unsigned long long)i;
            }
#endif
        }
    }
    for (int t = 0; t < TILES; t++) {   // v of the last index (what its bus launch would have written)
        const Dev& d = ((iters - 1) & 1) ? *dp1 : *dp0;
        const int tile = (t == 0) ? my_tile0 : my_tile1;
        if (TILES > 1 && tile < 0) break;
        const int slot = tile * BR_BLOCK + lane;
        if (slot >= d.L) continue;
        const int b = slot, base = 2 * d.Gp + b;
        const int4 bus4 = reinterpret_cast<const int4*>(d.br_bus4)[b];
        double rf[4] = {0.0, 0.0, 0.0, 0.0}, rt4[4] = {0.0, 0.0, 0.0, 0.0};
        if (PEERS) {
            const int ccode = rp.cut_code[b];
            const bool rem_f = ccode >= 0 && (ccode & 1) == 0, rem_t = ccode >= 0 && (ccode & 1) == 1;
            if (!rem_f) {
                if (!resident_wait(rs.ready + bus4.x, iters, rs.fail)) return;
                const double* pf = rs.busres[iters & 1] + 4 * (size_t)bus4.x;
#pragma unroll
                for (int m = 0; m < 4; m++) rf[m] = ld_agent(pf + m);
            }
            if (!rem_t) {
                if (!resident_wait(rs.ready + bus4.z, iters, rs.fail)) return;
                const double* pt = rs.busres[iters & 1] + 4 * (size_t)bus4.z;
#pragma unroll
                for (int m = 0; m < 4; m++) rt4[m] = ld_agent(pt + m);
            }
            {   // (every cut lane sends before any of them waits: see the loop's prologue)
                const int seq = (rp.epoch << 12) | iters;
                double out4[4], in4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < 4; m++) out4[m] = rem_f ? rt4[m] : rf[m];
                if (ccode >= 0) p2p_send4(rp.twin[ccode >> 1], seq, out4, rp.delay_ticks);
                bool ok = true;
                if (ccode >= 0) ok = p2p_wait4(rp.box + (size_t)P2P_ENTRY * (ccode >> 1), seq, rs.fail, in4, 5);
                if (!ok) return;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    rf[m] = rem_f ? in4[m] : rf[m];
                    rt4[m] = rem_t ? in4[m] : rt4[m];
                }
            }
        } else {
            if (!resident_wait2(rs.ready + bus4.x, rs.ready + bus4.z, iters, rs.fail)) return;
            const double* pf = rs.busres[iters & 1] + 4 * (size_t)bus4.x;
            const double* pt = rs.busres[iters & 1] + 4 * (size_t)bus4.z;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                rf[m] = ld_agent(pf + m);
                rt4[m] = ld_agent(pt + m);
            }
        }
        double uu[8], zz[8], ll[8], rr[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int e = base + k * Lp;
            uu[k] = d.u[e];
            zz[k] = d.z[e];
            ll[k] = d.l[e];
            rr[k] = d.rho[e];
        }
        const int hflag_raw = (TILES == 1 && HEAVY && nheavy > 0) ? rh.flag[b] : 0;   // (no adoption in the launch's last index: never fresh here)
        const int hflag = hflag_raw & 0xfffff;
        if (hflag != 0) {   // (the results of the last iteration from the heavy wavefront's hand-over, as in the loop)
            const double* const ho = rh.u + ((size_t)((iters - 1) & 1) * nheavy + (hflag - 1)) * 16;
#pragma unroll
            for (int k = 0; k < 8; k++) uu[k] = ld_agent(ho + k);
            if (((hflag_raw >> 21) & 1) != 0) adopted_results_to_planes(d, rs, ho, uu, b, base, d.Lp, rs.stamp0 + iters);
        }
        d.v[base] = (uu[0] + zz[0]) + (ll[0] + rf[0]) / rr[0];
        d.v[base + Lp] = (uu[1] + zz[1]) + (ll[1] + rf[1]) / rr[1];
        d.v[base + 2 * Lp] = (uu[2] + zz[2]) + (ll[2] + rt4[0]) / rr[2];
        d.v[base + 3 * Lp] = (uu[3] + zz[3]) + (ll[3] + rt4[1]) / rr[3];
        d.v[base + 4 * Lp] = rf[2];
        d.v[base + 5 * Lp] = rt4[2];
        d.v[base + 6 * Lp] = rf[3];
        d.v[base + 7 * Lp] = rt4[3];
    }
}

// Which branches get a heavy wavefront in the next resident launch (one block in front of it): those with the most solver trips in
// the last accepted x-updates (both buffer sets), at least 3, AND those that needed >= 3 trips in any of the last `window` resident
// iterations (last3: a thermal-limit-binding branch needs 4 - 5 trips in most iterations, not in all of them; picked from the last two
// iterations alone, a launch of 200 was paced by the stragglers that happened to rest when it started -- profiles/r05_heavy_wavefront_series.txt).
// At most `nmax` of them: all branches with key >= T, T the smallest threshold that fits; key = trips of the last two iterations, 3 for a
// branch that is only there by its history.  list[w] = the branch of heavy wavefront w (-1: none), flag[b] = w + 1 (0: the branch stays
// with its lane).  Which wavefront gets which branch is left to the order of the atomics: every branch is solved the same way wherever it runs.
// init[w][0..7] = z, [8..15] = l of that branch as the launch finds them: the branch's lane overwrites both in place with the z / l pass
// of the launch's first index, possibly before the heavy wavefront has looked (seen: ulp-level differences in a handful of branches).
// cut_code (a partition with peers, else nullptr): cut branches are passed by -- their lanes talk to the peers, a heavy wavefront never does.
__device__ __forceinline__ int heavy_key(const int* __restrict__ ticks0, const int* __restrict__ ticks1, const int* __restrict__ last3, int now, int window, int b)
{
    // trips of the last two iterations if >= 3; else by how long ago the branch last needed >= 3: within `window` iterations 3, within 4 x window 2,
    // within 16 x window 1 (the slots that the first tier leaves free go to the next one: the heavy set of a launch of 50 - 200 iterations is
    // better the more of the stragglers of the last few hundred iterations it holds -- measured with HONEST replays, see exaadmm_state_restore)
    const int k = min(63, max(ticks0[b], ticks1[b]));
    if (k >= 3) return k;
    const int h = last3[b];
    if (window <= 0 || h == 0) return 0;
    const int age = now - (h >> 4);
    return age <= window ? 3 : (age <= 4 * window ? 2 : (age <= 16 * window ? 1 : 0));
}
__global__ void __launch_bounds__(1024) k_heavy_select(const int* __restrict__ ticks0, const int* __restrict__ ticks1, int L, int Lp, int nmax,
                                                       int* __restrict__ list, int* __restrict__ flag, const double* __restrict__ z,
                                                       const double* __restrict__ l, int elem0, double* __restrict__ init, const int* __restrict__ cut_code,
                                                       const int* __restrict__ last3, int now, int window, int* __restrict__ req, int* __restrict__ cnt)
{
    // (one block: it sits between a call's opening launch and the resident launch, 30 us of every call when every thread walked its
    //  branches one load round trip after the other; now eight branches' loads are in flight per thread and the keys of the first pass wait
    //  in flag[] for the second)
    __shared__ int hist[64];
    __shared__ int thr, count;
    const int tid = threadIdx.x, nthreads = blockDim.x;   // (1 024 threads; 64 on request: one wavefront finds a free SIMD on a GPU that other
                                                           //  resident launches share, a block of 16 wavefronts needs a whole free CU -- tests with thread ranks)
    constexpr int U = 8;
    if (tid < 64) hist[tid] = 0;
    if (tid == 0) count = 0;
    __syncthreads();
    for (int b0 = tid; b0 < Lp; b0 += U * nthreads) {
        int key[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int b = b0 + j * nthreads;
            const bool on = b < L && (cut_code == nullptr || cut_code[b] < 0);
            key[j] = on ? heavy_key(ticks0, ticks1, last3, now, window, b) : 0;
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int b = b0 + j * nthreads;
            if (key[j] > 0) atomicAdd(&hist[key[j]], 1);
            if (b < Lp) flag[b] = key[j];
        }
    }
    for (int w = tid; w < nmax; w += nthreads) list[w] = -1;
    __syncthreads();
    if (tid == 0) {
        int t = 64, acc = 0;
        while (t > 1 && acc + hist[t - 1] <= nmax) {
            t--;
            acc += hist[t];
        }
        thr = t;   // (64: nobody)
    }
    __syncthreads();
    const int T = thr;
    for (int b0 = tid; b0 < Lp; b0 += U * nthreads) {
        int key[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int b = b0 + j * nthreads;
            key[j] = (b < Lp) ? flag[b] : 0;   // (written by this thread above)
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int b = b0 + j * nthreads;
            int f = 0;
            if (key[j] >= T && key[j] > 0) {
                const int pos = atomicAdd(&count, 1);
                list[pos] = b;
                f = pos + 1;
            }
            if (b < Lp) flag[b] = f;
        }
    }
    __syncthreads();
    const int n = count;
    for (int q = tid; q < 16 * n; q += nthreads) {
        const int w = q >> 4, k = q & 7, e = elem0 + list[w] + k * Lp;
        init[q] = (q & 8) ? l[e] : z[e];
    }
    // adoption (ResidentHeavy): no request yet, no ticket taken, the free heavy wavefronts start behind the n chosen here
    if (req != nullptr) {
        for (int w = tid; w < nmax; w += nthreads) req[w] = 0;
        if (tid == 0) {
            cnt[0] = n;
            cnt[1] = 0;
        }
    }
}

// (a history like k_heavy_select's was tried for these keys -- a tile whose straggler rests when the launch starts keeps its rank --: 8.08 k against
//  8.35 k it/s at 88 207 branches, whole solve + 0.7 %: not kept)
__global__ void __launch_bounds__(256) k_tile_keys(const int* __restrict__ ticks0, const int* __restrict__ ticks1, int L, int nb, int* __restrict__ key,
                                                   const int* __restrict__ cut_code)
{
    const int t = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= nb) return;
    const int b = t * BR_BLOCK + lane;
    int k = (b < L) ? max(ticks0[b], ticks1[b]) : 0;
    if (cut_code != nullptr && b < L && cut_code[b] >= 0) k |= 0x100000;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k = max(k, __shfl_down(k, off, 64));
    if (lane == 0) key[t] = k;
}

// order[0 .. nb) = the tiles by falling key (k_tile_keys: the largest trip count of a tile's lanes in the last two iterations); which of two tiles
// with the same key comes first is left to the atomics -- a tile is solved the same way whichever block solves it (Dev::blk_order)
__global__ void __launch_bounds__(1024) k_block_order(const int* __restrict__ key, int nb, int* __restrict__ order)
{
    __shared__ int hist[64], start[64];
    const int tid = threadIdx.x;
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < nb; t += blockDim.x) atomicAdd(&hist[min(63, key[t] & 0xfffff)], 1);
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int k = 63; k >= 0; k--) {
            start[k] = pos;
            pos += hist[k];
        }
    }
    __syncthreads();
    for (int t = tid; t < nb; t += blockDim.x) order[atomicAdd(&start[min(63, key[t] & 0xfffff)], 1)] = t;
}

__global__ void __launch_bounds__(1024) k_tile_assign(const int* __restrict__ key_g, int nb, int wb, int* __restrict__ tiles)
{
    extern __shared__ int sh[];   // key[nb], order[nb]
    int* key = sh;
    int* order = sh + nb;
    for (int t = threadIdx.x; t < nb; t += blockDim.x) key[t] = key_g[t];
    __syncthreads();
    for (int t = threadIdx.x; t < nb; t += blockDim.x) {
        const int k = key[t];
        int rank = 0;
        for (int s = 0; s < nb; s++) rank += (key[s] > k || (key[s] == k && s < t)) ? 1 : 0;
        order[rank] = t;
    }
    __syncthreads();
    const int singles = 2 * wb - nb, pairs = nb - wb;
    for (int w = threadIdx.x; w < wb; w += blockDim.x) {
        if (w < singles) {
            tiles[2 * w] = order[w];
            tiles[2 * w + 1] = -1;
        } else {
            const int k = w - singles;
            tiles[2 * w] = order[singles + k];
            tiles[2 * w + 1] = (k < pairs) ? order[nb - 1 - k] : -1;
        }
    }
}

// The receiving half of a partition's one exchange per iteration (dist_native.h), after the grouped ncclSend / ncclRecv:
//   * thread i scatters the 4 v entries of cut branch i that the peer's bus update produced (the half of the branch
//     that lives on the peer's bus) into v;
//   * thread 0 adds the ranks' partial residual sums IN RANK ORDER (the same order, hence the same bits and the same
//     decision, on every rank) and advances the loop control -- or, out != nullptr, just delivers the totals.
// rcode[i] = (side << 28) | local branch, side END_FROM / END_TO = which half ARRIVES, rpos[i] = where in the receive
// buffer (a peer's message is its cut branches' entries followed by its four sums); n_cut = 0: sums only.
// sums_mode: 0 = the message carries sums: add them; 1 = only if ctrl->pending says so (serial callers); 2 = no sums in this message
__global__ void k_halo_unpack(Dev d, const int* rcode, const int* rpos, int n_cut, Ctrl* ctrl, double* out, int sums_mode)
{
    if (ctrl != nullptr && ctrl->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cut) {
        const int code = rcode[i], c = code & END_INDEX;
        const bool from = ((code >> 28) & 3) == END_FROM;
        const double* q = d.halo_recv + rpos[i];
        const int e = 2 * d.Gp + c, Lp = d.Lp;
        d.v[e + (from ? 0 : 2) * Lp] = q[0];
        d.v[e + (from ? 1 : 3) * Lp] = q[1];
        d.v[e + (from ? 4 : 5) * Lp] = q[2];
        d.v[e + (from ? 6 : 7) * Lp] = q[3];
    }
    if (i != 0) return;
    if (sums_mode == 2 || (sums_mode == 1 && !ctrl->pending)) return;   // first iteration of a loop: no residual sums travel yet
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = 0; p < d.nranks; p++) {
        const double* q = (p == d.rank) ? d.halo_local : d.halo_recv + d.halo_off[d.nranks + p];
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] += q[j];
    }
    if (out != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; j++) out[j] = s[j];
    }
    if (ctrl != nullptr) ctrl_advance(ctrl, s);
}

// sums-only exchange (end of an inner loop, ||z|| of the outer loop): this rank's four sums into every peer's message
__global__ void k_halo_pack_sums(Dev d, const double* sums4, const Ctrl* ctrl)
{
    if (loop_done(ctrl)) return;
    const int j = threadIdx.x;
    if (j >= 4) return;
    d.halo_local[j] = sums4[j];
    for (int p = 0; p < d.nranks; p++)
        if (p != d.rank) d.halo_send[d.halo_off[p] + j] = sums4[j];
}

// ---------------------------------------------------------------- elementwise, step-wise
__global__ void k_copy_z(Dev d)  // copy_data_kernel (utilities_gpu.jl:1-8)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) d.zp[i] = d.z[i];
}

// Element i of the SoA buffers counts in this partition's share of the norms: everything except the replicas of a
// peer's cut branches (stored after the Lown owned ones in every branch plane; single GPU: Lown = L).  The updates
// themselves run on every element -- replicas are kept bit-identical on both owners.
__device__ __forceinline__ bool elem_counts(const Dev& d, int i)
{
    return d.Lown >= d.L || i < 2 * d.Gp || (i - 2 * d.Gp) % d.Lp < d.Lown;
}

__global__ void k_update_z(Dev d, double beta)  // update_zv_kernel (acopf_admm_update_z_gpu.jl:1-11)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        d.z[i] = (-(d.lz[i] + d.l[i] + d.rho[i] * (d.u[i] - d.v[i]))) / (beta + d.rho[i]);
    }
}

__global__ void k_update_l(Dev d, double beta)  // update_l_kernel (acopf_admm_update_l_gpu.jl:1-14)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        d.l[i] = -(d.lz[i] + beta * d.z[i]);
    }
}

__global__ void k_update_lz(Dev d, double beta, double maxmult)  // update_lz_kernel (acopf_admm_update_lz_gpu.jl:1-9)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        d.lz[i] = fmax(-maxmult, fmin(maxmult, d.lz[i] + (beta * d.z[i])));
    }
}

// wavefront + block reduction of 4 running sums; thread 0 of the block gets the totals.
// Fixed topology -> bitwise reproducible for a given launch shape.
__device__ __forceinline__ void block_reduce4(double (&s)[4])
{
    __shared__ double sh[4][EW_BLOCK / 64];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) sh[j][wave] = s[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double t = 0.0;
            for (int w = 0; w < EW_BLOCK / 64; w++) t += sh[j][w];
            s[j] = t;
        }
    }
}

// compute_primal_residual_kernel + 2x vector_difference + 4 norms
// (acopf_admm_update_residual_gpu.jl:1-29) in one pass; materialises rp, rd, Ax_plus_By.
__global__ void __launch_bounds__(EW_BLOCK) k_residual(Dev d)
{
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        const double z = d.z[i];
        const double rp = d.u[i] - d.v[i] + z;
        const double rd = z - d.zp[i];
        const double ax = rp - z;
        d.rp[i] = rp;
        d.rd[i] = rd;
        d.ax[i] = ax;
        if (!elem_counts(d, i)) continue;
        s[0] += rp * rp;
        s[1] += rd * rd;
        s[2] += z * z;
        s[3] += ax * ax;
    }
    block_reduce4(s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) d.partials[blockIdx.x * 4 + j] = s[j];
    }
}

__global__ void __launch_bounds__(EW_BLOCK) k_sumsq_z(Dev d)
{
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        if (!elem_counts(d, i)) continue;
        const double z = d.z[i];
        s[2] += z * z;
    }
    block_reduce4(s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) d.partials[blockIdx.x * 4 + j] = s[j];
    }
}

// ---------------------------------------------------------------- elementwise, fused
// z_prev <- z ; z-update ; l-update ; residual sums, one pass, 6 reads + 3 writes per element.
// (admm_inner_prestep + admm_update_z + admm_update_l + admm_update_residual of the reference:
//  7 launches + 4 library norms there.)
__global__ void __launch_bounds__(EW_BLOCK) k_fused_zl(Dev d, double beta_arg, const Ctrl* ctrl, int* mark_pending)
{
    if (loop_done(ctrl)) return;
    if (mark_pending != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *mark_pending = 1;   // read by the NEXT launch only
    const double beta = ctrl ? ctrl->beta : beta_arg;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        const double u = d.u[i], v = d.v[i], rho = d.rho[i], lz = d.lz[i];
        const double zold = (d.z_in ? d.z_in : d.z)[i];   // (three-set rotation of a partition with peers: Dev::z_in)
        const double z = (-(lz + (d.l_in ? d.l_in : d.l)[i] + rho * (u - v))) / (beta + rho);
        const double rp = u - v + z;
        const double rd = z - zold;
        const double ax = rp - z;
        d.zp[i] = zold;
        d.z[i] = z;
        d.l[i] = -(lz + beta * z);
        if (!elem_counts(d, i)) continue;
        s[0] += rp * rp;
        s[1] += rd * rd;
        s[2] += z * z;
        s[3] += ax * ax;
    }
    block_reduce4(s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) d.partials[blockIdx.x * 4 + j] = s[j];
    }
}

// Sum the per-block partials in a fixed order; optionally advance the loop control:
// inner++, res = sqrt(sums), done = (primres <= eps_pri) || inner == max_inner
// (admm_two_level.jl:45,60-62 evaluated on the device).  advance = 0: the sums only (a partition with peers adds the
// other ranks' sums before anything is decided: k_halo_pack_sums / k_halo_unpack).
__global__ void __launch_bounds__(EW_BLOCK) k_final_reduce(const double* partials, int nblk, double* out_sums, Ctrl* ctrl, int need_pending,
                                                           int advance)
{
    if (ctrl != nullptr && ctrl->done) return;
    if (need_pending && !ctrl->pending) return;   // speculative loop: the bus launch already reduced this iteration
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += EW_BLOCK) {
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] += partials[b * 4 + j];
    }
    block_reduce4(s);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) out_sums[j] = s[j];
        if (ctrl != nullptr && advance) ctrl_advance(ctrl, s);
    }
}

// ---------------------------------------------------------------- multi-period elementwise steps
// ModelMpacopf.  The residual norms of the multi-period model are the MAXIMUM over
// periods of the per-period norms, period t >= 1 including its ramp block
// (mpacopf_admm_update_residual_cpu.jl:1-55), so the elementwise kernels are launched on a
// (chunk, period, plane) grid: plane 0..9 = the SoA planes, plane 10 = the ramp block, whose consensus
// partner is v[pg] of the PREVIOUS period (mpacopf_admm_update_z_cpu.jl:10-14).  Each block stays
// inside one period, partial sums are kept per (period, plane, chunk) and reduced in a fixed order.
constexpr int MP_PLANES = 11;
enum { MP_COPY_Z = 0, MP_UPDATE_Z = 1, MP_UPDATE_L = 2, MP_UPDATE_LZ = 3, MP_RESIDUAL = 4, MP_SUMSQ_Z = 5, MP_FUSED_ZL = 6 };

template <int MODE>
__global__ void __launch_bounds__(EW_BLOCK) k_mp_elem(Dev d, double beta_arg, double maxmult, const Ctrl* ctrl)
{
    if (loop_done(ctrl)) return;
    const double beta = ctrl ? ctrl->beta : beta_arg;
    const int plane = blockIdx.z, t = blockIdx.y;
    const bool ramp = (plane == MP_PLANES - 1);
    const int n = (plane < 2 || ramp) ? d.g1 : d.l1;
    const int off = ramp ? t * d.g1 : ((plane < 2 ? plane * d.Gp : 2 * d.Gp + (plane - 2) * d.Lp) + t * n);
    const int voff = ramp ? (t - 1) * d.g1 : off;
    double* const U = ramp ? d.r_u : d.u;
    double* const Z = ramp ? d.r_z : d.z;
    double* const ZP = ramp ? d.r_zp : d.zp;
    double* const Lm = ramp ? d.r_l : d.l;
    double* const LZ = ramp ? d.r_lz : d.lz;
    const double* const RHO = ramp ? d.r_rho : d.rho;
    double* const RP = ramp ? d.r_rp : d.rp;
    double* const RD = ramp ? d.r_rd : d.rd;
    double* const AX = ramp ? d.r_ax : d.ax;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (!(ramp && t == 0)) {
        for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gridDim.x * EW_BLOCK) {
            const int e = off + i, ev = voff + i;
            if (MODE == MP_COPY_Z) {
                ZP[e] = Z[e];
            } else if (MODE == MP_UPDATE_Z) {
                Z[e] = (-(LZ[e] + Lm[e] + RHO[e] * (U[e] - d.v[ev]))) / (beta + RHO[e]);
            } else if (MODE == MP_UPDATE_L) {
                Lm[e] = -(LZ[e] + beta * Z[e]);
            } else if (MODE == MP_UPDATE_LZ) {
                LZ[e] = fmax(-maxmult, fmin(maxmult, LZ[e] + (beta * Z[e])));
            } else if (MODE == MP_RESIDUAL) {
                const double z = Z[e];
                const double rp = U[e] - d.v[ev] + z;
                const double rd = z - ZP[e];
                const double ax = rp - z;
                RP[e] = rp;
                RD[e] = rd;
                AX[e] = ax;
                s[0] += rp * rp;
                s[1] += rd * rd;
                s[2] += z * z;
                s[3] += ax * ax;
            } else if (MODE == MP_SUMSQ_Z) {
                const double z = Z[e];
                s[2] += z * z;
            } else {  // MP_FUSED_ZL: z_prev <- z, z-update, l-update, residual sums in one pass
                const double u = U[e], v = d.v[ev], rho = RHO[e], lz = LZ[e];
                const double zold = Z[e];
                const double z = (-(lz + Lm[e] + rho * (u - v))) / (beta + rho);
                const double rp = u - v + z;
                const double rd = z - zold;
                const double ax = rp - z;
                ZP[e] = zold;
                Z[e] = z;
                Lm[e] = -(lz + beta * z);
                s[0] += rp * rp;
                s[1] += rd * rd;
                s[2] += z * z;
                s[3] += ax * ax;
            }
        }
    }
    if (MODE == MP_RESIDUAL || MODE == MP_SUMSQ_Z || MODE == MP_FUSED_ZL) {
        block_reduce4(s);
        if (threadIdx.x == 0) {
            const int slot = (t * MP_PLANES + plane) * gridDim.x + blockIdx.x;
#pragma unroll
            for (int j = 0; j < 4; j++) d.partials[slot * 4 + j] = s[j];
        }
    }
}

// Per-period sums in a fixed order, norm = sqrt, model norm = max over periods; optionally advance the
// loop control exactly like k_final_reduce.  out[j] receives the NORMS (not their squares).
__global__ void __launch_bounds__(EW_BLOCK) k_final_reduce_mp(const double* partials, int nblk_per_period, int T, double* out,
                                                              Ctrl* ctrl)
{
    if (ctrl != nullptr && ctrl->done) return;
    double best[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = 0; t < T; t++) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        const double* p = partials + (size_t)t * nblk_per_period * 4;
        for (int b = threadIdx.x; b < nblk_per_period; b += EW_BLOCK) {
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] += p[b * 4 + j];
        }
        block_reduce4(s);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) best[j] = fmax(best[j], sqrt(s[j]));
        }
        __syncthreads();   // block_reduce4's scratch is reused by the next period
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) out[j] = best[j];
        if (ctrl != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ctrl->sums[j] = best[j] * best[j];
                ctrl->res[j] = best[j];
            }
            const int inner = ctrl->inner + 1;
            ctrl->inner = inner;
            const bool test = (ctrl->check_every <= 1) || (inner % ctrl->check_every == 0);
            if ((test && ctrl->res[0] <= ctrl->eps_pri) || inner >= ctrl->max_inner) ctrl->done = 1;
        }
    }
}

// Python-driven distributed loop: consume all-reduced sums (admm_two_level.jl:45,60-62 on the device)
__global__ void k_ctrl_step(const double* sums, Ctrl* ctrl)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || ctrl->done) return;
    const double s[4] = {sums[0], sums[1], sums[2], sums[3]};
    ctrl_advance(ctrl, s);
}

// rp, rd, Ax_plus_By from u, v, z, z_prev (only needed when a caller downloads them after a fused loop)
__global__ void k_materialize(Dev d)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.N; i += gridDim.x * blockDim.x) {
        const double z = d.z[i];
        const double rp = d.u[i] - d.v[i] + z;
        d.rp[i] = rp;
        d.rd[i] = z - d.zp[i];
        d.ax[i] = rp - z;
    }
}

// halo exchange helpers: gather / scatter (vector id, SoA index) lists
struct VecTable {
    double* p[10];
};

__global__ void k_gather(VecTable t, const int* which, const int* idx, int n, double* dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = t.p[which[i]][idx[i]];
}

__global__ void k_scatter(VecTable t, const int* which, const int* idx, int n, const double* src)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t.p[which[i]][idx[i]] = src[i];
}

}  // namespace exa
