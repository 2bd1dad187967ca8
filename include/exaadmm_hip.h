/*
 * exaadmm_hip.h -- C ABI of the MI355X-native backend for ExaAdmm.jl's two-level
 * ADMM ACOPF inner loop (libexaadmm_hip.so, HIP for gfx950).
 *
 * Drop-in boundary: the reference drives its backends through 12 generic functions
 * dispatched on (env::AdmmEnv, mod::AbstractOPFModel, device)
 *   /root/reference/docs/src/dev.md:65-155, called from
 *   /root/reference/src/algorithms/admm_two_level.jl:29-81.
 * A Julia method `admm_update_x(env, mod, dev::HipNative) = ccall((:exaadmm_update_x, lib), ...)`
 * per step binds to exactly the entry points below (see INTEGRATION.md and
 * julia/ExaAdmmHip.jl).  All functions are extern "C", take plain pointers and sizes,
 * return 0 on success or a negative exaadmm_status; no exception crosses the boundary.
 *
 * Conventions
 *  - The library owns all device memory behind the opaque handle; callers own the host
 *    buffers they pass.  Host vectors use the REFERENCE layout: length
 *    nvar = 2*ngen + 8*nline, generators first [pg,qg], then per branch
 *    [pij,qij,pji,qji,wi,wj,ti,tj] (src/models/acopf/acopf_model.jl:55-58).  Device
 *    storage is struct-of-arrays; conversion happens inside upload/download.
 *  - Index arrays are int64 as Julia holds them; exaadmm_grid.index_base says whether
 *    they are 1-based (Julia) or 0-based (C/Python).
 *  - Step-wise entry points are synchronous on return (the reference reads mod.info right
 *    after each step).  The fused entry points keep control on the device side.
 *  - Not re-entrant per handle; one handle per GPU, different handles may be driven
 *    from different host threads.
 */
#ifndef EXAADMM_HIP_H
#define EXAADMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct exaadmm_handle exaadmm_handle;
typedef struct exaadmm_case exaadmm_case;

typedef enum {
    EXAADMM_OK = 0,
    EXAADMM_ERR_INVALID = -1,   /* bad argument / inconsistent sizes */
    EXAADMM_ERR_HIP = -2,       /* HIP runtime error (message in exaadmm_last_error) */
    EXAADMM_ERR_NO_DEVICE = -3, /* no usable gfx950 device: there is NO CPU fallback */
    EXAADMM_ERR_IO = -4,        /* file not found / parse error */
    EXAADMM_ERR_ALLOC = -5
} exaadmm_status;

/* Mirrors GridData (src/utils/grid_data.jl:3-83), fields the ACOPF path reads. */
typedef struct {
    int64_t ngen, nline, nbus;
    int32_t index_base; /* 1: arrays exactly as Julia holds them; 0: zero-based */
    double baseMVA;
    const double *pgmin, *pgmax, *qgmin, *qgmax, *c2, *c1, *c0;           /* ngen */
    const double *YffR, *YffI, *YftR, *YftI, *YttR, *YttI, *YtfR, *YtfI;  /* nline */
    const double *FrVmBound, *ToVmBound, *FrVaBound, *ToVaBound;          /* 2*nline (lo,hi) */
    const double *rateA;                                                  /* nline */
    const int64_t *brBusIdx;                                              /* 2*nline (from,to) */
    const double *Pd, *Qd, *Vmin, *Vmax, *YshR, *YshI;                    /* nbus */
    const int64_t *FrStart, *FrIdx, *ToStart, *ToIdx, *GenStart, *GenIdx; /* CSR bus -> branch/gen */
    /* partitions only (multi-GPU): the first nline_owned branches are this rank's own (their from-bus is
     * one of its buses) and count in the residual norms; the rest are replicas of cut branches a peer owns
     * (to-bus local), solved and updated here as well, bit-identically.  0 = everything is owned (single GPU) */
    int64_t nline_owned;
} exaadmm_grid;

/* Mirrors Parameters (src/utils/environment.jl:6-76), fields the path reads. */
typedef struct {
    double mu_max;         /* 1e8 */
    int32_t max_auglag;    /* 50 */
    double initial_beta;   /* 1e3 */
    double beta;           /* running value */
    double inc_c;          /* 6 */
    double theta;          /* 0.8 */
    double outer_eps;      /* 2e-4 */
    double MAX_MULTIPLIER; /* 1e12 */
    int32_t outer_iterlim; /* 20 */
    int32_t inner_iterlim; /* 1000 */
    double scale;          /* 1e-4 */
    int32_t verbose;       /* 0 */
    int32_t check_every;   /* 1 = exact per-iteration inner-loop break (reference semantics);
                              k>1 = opt-in relaxed mode: test convergence every k iterations */
} exaadmm_params;

/* Mirrors IterationInformation (src/utils/environment.jl:328-358). */
typedef struct {
    int32_t status; /* 0 = :IterationLimit, 1 = :Solved */
    int32_t inner, outer, cumul;
    double objval, primres, dualres, mismatch, norm_z_curr, norm_z_prev, eps_pri;
    double time_overall;                                    /* seconds, scope of admm_two_level.jl:28-80 */
    double time_x_update, time_xbar_update, time_zl_update; /* device time from HIP events (fused: sampled) */
    double avg_auglag_it, avg_minor_it;                     /* of the last x-update (cpu path returns these) */
} exaadmm_info;

/* which vector of Solution (src/utils/environment.jl:177-226) */
typedef enum {
    EXAADMM_U_CURR = 0, EXAADMM_V_CURR = 1, EXAADMM_L_CURR = 2, EXAADMM_RHO = 3,
    EXAADMM_RD = 4, EXAADMM_RP = 5, EXAADMM_Z_CURR = 6, EXAADMM_Z_PREV = 7,
    EXAADMM_LZ = 8, EXAADMM_AX_PLUS_BY = 9,
    /* ramp block of the multi-period model only (SolutionRamping.s_curr, gen_membuf rows 7 and 8) */
    EXAADMM_RAMP_S = 20, EXAADMM_RAMP_MU = 21, EXAADMM_RAMP_XI = 22
} exaadmm_vector;

const char *exaadmm_version(void);
const char *exaadmm_last_error(void);
int exaadmm_device_count(void); /* visible HIP devices (0 = none); starts the library's HIP runtime */
/* text: path and version of the HIP runtime the library is bound to in this process, and the HIP version it was
 * compiled against; returns (runtime major - compiled major), 0 = compatible */
int exaadmm_hip_runtime_info(char *out, int64_t n);
const char *exaadmm_io_error(void); /* message of the last failed exaadmm_load_matpower */
void exaadmm_default_params(exaadmm_params *p);

/* ModelAcopf(env) without init: uploads the grid (src/models/acopf/acopf_model.jl:41-94) */
int exaadmm_create(const exaadmm_grid *g, const exaadmm_params *p, int device_id, exaadmm_handle **out);
int exaadmm_destroy(exaadmm_handle *h);
int exaadmm_set_params(exaadmm_handle *h, const exaadmm_params *p);
int exaadmm_get_params(exaadmm_handle *h, exaadmm_params *p);
int64_t exaadmm_nvar(exaadmm_handle *h);

/* init_solution! (acopf_init_solution_gpu.jl:1-64): resets the Solution vectors (and the ramp blocks of a
 * multi-period handle) only.  membuf (line-limit multipliers / penalty, rateA) and pgmin_curr / pgmax_curr
 * are filled by exaadmm_create, like the ModelAcopf constructor does (acopf_model.jl:61-64,87-89), and
 * survive this call -- solve_mpacopf's warm start and the rolling horizon depend on that. */
int exaadmm_init_solution(exaadmm_handle *h, double rho_pq, double rho_va);

/* host <-> device, reference layout */
int exaadmm_upload_solution(exaadmm_handle *h, int which, const double *host, int64_t n);
int exaadmm_download_solution(exaadmm_handle *h, int which, double *host, int64_t n);
/* membuf is 31 x nline column-major as in the reference; rows 25,26,27 (line-limit
 * multipliers, penalty) and 29 (rateA) are state, rows 1-24 are scratch the device keeps
 * in registers/LDS and returns as zeros (acopf_auglag_linelimit_kernel_gpu.jl:49-79). */
int exaadmm_upload_membuf(exaadmm_handle *h, const double *host, int64_t nrow, int64_t nline);
int exaadmm_download_membuf(exaadmm_handle *h, double *host, int64_t nrow, int64_t nline);
/* rolling-horizon hooks (acopf_admm_rolling_gpu.jl:1-14): per-period load and ramp-limited bounds */
int exaadmm_set_load(exaadmm_handle *h, const double *Pd, const double *Qd, int64_t nbus);
int exaadmm_set_pg_bounds(exaadmm_handle *h, const double *pgmin_curr, const double *pgmax_curr, int64_t ngen);

/* ---- one entry point per plugin step (docs/src/dev.md:65-155) ---- */
int exaadmm_inner_prestep(exaadmm_handle *h);            /* acopf_admm_prepoststep_gpu.jl:11-21  */
int exaadmm_update_x(exaadmm_handle *h, int major_iter); /* acopf_admm_update_x_gpu.jl:48-57     */
int exaadmm_update_x_gen(exaadmm_handle *h);             /* acopf_generator_kernel_gpu.jl:1-33   */
int exaadmm_update_x_line(exaadmm_handle *h, int major_iter); /* acopf_auglag_linelimit_kernel_gpu.jl */
int exaadmm_update_xbar(exaadmm_handle *h);              /* acopf_bus_kernel_gpu.jl:1-119        */
int exaadmm_update_z(exaadmm_handle *h, double beta);    /* acopf_admm_update_z_gpu.jl:1-24      */
int exaadmm_update_l(exaadmm_handle *h, double beta);    /* acopf_admm_update_l_gpu.jl:1-26      */
/* out = {primres, dualres, norm_z_curr, mismatch}         acopf_admm_update_residual_gpu.jl:11-29 */
int exaadmm_update_residual(exaadmm_handle *h, double out[4]);
int exaadmm_update_lz(exaadmm_handle *h, double beta);   /* acopf_admm_update_lz_gpu.jl:1-21     */
int exaadmm_norm_z(exaadmm_handle *h, double *out);      /* admm_outer_prestep: ||z_curr||       */
int exaadmm_objective(exaadmm_handle *h, double *out);   /* admm_poststep objective              */
int exaadmm_branch_stats(exaadmm_handle *h, double *avg_auglag_it, double *avg_minor_it);
/* per-branch work counters of the last x-update: augmented-Lagrangian iterations, TRON minor
 * iterations (summed), solver-loop trips of the lane (divergence diagnostics) */
int exaadmm_download_branch_counters(exaadmm_handle *h, int32_t *auglag_it, int32_t *minor_it, int32_t *ticks, int64_t nline);
/* per tile of 64 branches in the device's storage order (= per wavefront of the branch launch): the largest solver-trip count of its
 * lanes in the last x-update; ntiles = ceil(nline / 64) */
int exaadmm_download_tile_trips(exaadmm_handle *h, int32_t *out, int64_t ntiles);
/* resident wavefronts per SIMD that the branch kernel's register / LDS footprint allows (occupancy API) */
int exaadmm_branch_kernel_occupancy(exaadmm_handle *h, int *waves_per_simd);
/* which form of the inner loop exaadmm_run_inner / _run_fixed / _solve_two_level use on this handle (diagnostic):
 * 0 = four launches per iteration, 1 = two launches per iteration (speculative loop), 2 = resident loop (one launch per
 * batch of iterations, every wavefront resident; option "resident" = 0 selects 1) */
int exaadmm_loop_kind(exaadmm_handle *h, int *kind);
/* The resident loop needs every wavefront of its launch on the chip at once.  When that turns out not to hold (the GPU is
 * shared with other work: a capped wait inside the launch reports it, or the runtime refuses the cooperative launch) the
 * call restores the iterate it started from, repeats its loop with two launches per iteration and returns what it would
 * have returned -- the reference's loop cannot fail there either (admm_two_level.jl:34-63); the handle then stays on
 * that loop.  count = how often that has happened on this handle (0 or 1). */
int exaadmm_loop_downgrades(exaadmm_handle *h, int *count);
/* What the LAST loop call of this handle (exaadmm_run_inner / _run_fixed / _solve_two_level) actually launched, as the library
 * decided it (occupancy figures of this device, co-residency, options) -- reporting tools ask instead of re-deriving the rule:
 * tiles = tiles of 64 branches per resident wavefront (0: the call did not run the resident loop, 1, 2); heavy = heavy wavefronts
 * in the call's last resident launch (0: none); launches = resident launches of the call; wavefronts = grid of the last one. */
int exaadmm_resident_form(exaadmm_handle *h, int *tiles, int *heavy, int *launches, int *wavefronts);
/* First-use self check of the resident loops (admm_two_level.jl:34-63: the loop must not return iterates the reference's loop
 * cannot).  The cross-wavefront hand-overs of k_resident rest on hardware behaviour, so the first resident call of a handle -- and
 * the first one with heavy wavefronts -- is preceded by a CHECK: from the iterate the call starts with, 8 iterations are run with the
 * resident loop and with the two-launch loop, and u, v, z, z_prev, l, the line-limit multipliers / penalty, the branch counters and
 * the four norms are compared bit for bit on the device; the iterate is restored either way.  A difference downgrades the handle
 * to the two-launch loop (exaadmm_loop_downgrades counts it, exaadmm_last_error says what differed) before the call proper runs.
 * A form that has passed on this device for a grid of this size in this process is not checked again for another handle of the same
 * size (the check is about the chip's treatment of that kernel at that launch geometry, not about the handle).
 * checks = self checks this handle made so far, mismatches = how many of them found a difference.  Option "trust_resident" = 1 skips them. */
int exaadmm_resident_selfcheck_stats(exaadmm_handle *h, int *checks, int *mismatches);

/* ---- options: A/B switches and diagnostics of a handle ----
 * Every option has a default and none is needed in normal use; the library reads NO environment variable of its own (hosts that
 * want switches from the environment translate them: exaadmm.jl_amd/__init__.py, tools/exaadmm_solve.c).  Set after create,
 * between loop calls.  Keys (default):
 *   "bus_kernel_lane" (0)  1: one lane per bus, the reference-order twin k_bus; implies the four-launch loop
 *   "graph" (1)            batches of the two- / four-launch loops (a partition with peers: incl. their exchanges) replayed from a hipGraph
 *   "spec" (1)             the speculative two-launch loop (0: four launches per iteration); single- and multi-period handles alike
 *   "rec" (1)              pushed bus records in the fused loops (0: gathers like the step-wise path; no resident loop then)
 *   "resident" (1)         the resident loop where the grid fits the chip
 *   "resident_tiles" (1)   its two-tiles-per-wavefront form for grids up to twice the chip's SIMD count
 *   "heavy" (1)            heavy wavefronts (the branches with the most solver trips on wavefronts of their own)
 *   "heavy_max" (512)      at most this many of them (they take the SIMDs the launch leaves idle; fewer when the GPU is shared)
 *   "heavy_adopt" (4)      a branch lane whose solve took at least this many trips hands its branch to one of the launch's free heavy wavefronts, from the
 *                          next iteration on (0: the heavy wavefronts of a launch are those chosen in front of it)
 *   "heavy_split" (48)     > 0: the first launch of a convergence-tested inner loop ends after this many iterations and the rest of the loop runs with heavy
 *                          wavefronts -- when the handle's previous inner loop ran long enough to promise the extra launch back; 0: one launch per 1 024 iterations
 *   "heavy_split_prev" (96) ... at least this many iterations (a handle's first loop is always split)
 *   "block_order" (1)      launches with more blocks than the chip has SIMDs (the multi-period model, grids beyond the resident loop): the generator / ramp
 *                          blocks first, then the tiles of 64 branches by falling solver-trip count (0: in storage order)
 *   "dist_overlap" (1)     partition with peers, two launches per iteration: the exchange on a second stream beside the interior solves
 *   "dist_p2p" (1)         partition with peers whose mailboxes are mapped (exaadmm_dist_p2p_import): the resident loop
 *   "trust_resident" (0)   1: no first-use self check of the resident loops
 * (Fifteen keys.  What rounds 4-5 also offered: heavy_from / heavy_window are constants, resident_chunk / heavy_select_small are gone, mp_spec is
 *  "spec", dist_graph is "graph", resident_coop / resident_cap_limit are switches of the test build's exaadmm_test_hook only, and downgrades are
 *  reported on stderr when exaadmm_params.verbose > 0.)
 * Unknown key: EXAADMM_ERR_INVALID. */
int exaadmm_set_option(exaadmm_handle *h, const char *key, int64_t value);
int exaadmm_get_option(exaadmm_handle *h, const char *key, int64_t *value);

/* ---- fused: control stays on the device side ---- */
/* Inner while-loop of admm_two_level (admm_two_level.jl:34-63) for outer iteration `outer`:
 * runs until primres <= sqrt(nvar)/(2500*outer) or max_inner iterations.
 * inner_done = iterations executed; out = the four norms of the last one. */
int exaadmm_run_inner(exaadmm_handle *h, int outer, int max_inner, double beta, int *inner_done, double out[4]);
/* Whole admm_two_level (admm_two_level.jl:1-88) with the handle's parameters. */
int exaadmm_solve_two_level(exaadmm_handle *h, exaadmm_info *out);
/* K inner iterations with no convergence test (benchmark / roofline mode; counts as major_iter
 * first_major_iter, first_major_iter+1, ...).  If ms_branch != NULL it receives the mean
 * duration of the branch kernel measured with HIP events on the launch stream. */
int exaadmm_run_fixed(exaadmm_handle *h, int iters, int first_major_iter, double beta, double out[4],
                      double *ms_total, double *ms_branch);

/* device-side snapshot / restore of the iterate (u, v, z, z_prev, l, lz, line-limit multipliers and penalty; on a multi-period
 * handle the ramp blocks as well) AND of what the resident launches choose their heavy wavefronts / tile pairs from (the per-branch
 * solver-trip counters of every buffer set, the trip history and its clock): benchmark drivers time repeated stretches of iterations
 * from one defined state -- a repeat must not start with knowledge of what the previous repeat saw */
int exaadmm_state_save(exaadmm_handle *h);
int exaadmm_state_restore(exaadmm_handle *h);

/* ---- multi-GPU (one handle = one partition of the grid, one process per GPU) ----
 * Buses are partitioned; a partition holds the generators and branches of its buses.  A branch whose two buses
 * belong to different partitions (a cut branch) is held and SOLVED by both: its u, z, l, lz and line-limit
 * multipliers are bit-identical replicas, and the only thing a partition must be told per iteration is the half
 * of such a branch's v that the peer's bus update produces (4 doubles per cut branch, each way).
 * exaadmm_create flags the cut branches itself: a branch listed in only one of FrIdx / ToIdx has its other end on
 * a peer's bus.
 *
 * In-library loop (RCCL over xGMI, bound at run time): once a handle is attached to its peers, the ordinary entry
 * points exaadmm_run_inner / exaadmm_run_fixed / exaadmm_norm_z / exaadmm_solve_two_level are collective over the
 * ranks: per inner iteration ONE grouped ncclSend / ncclRecv per peer (the cut branches' new v entries + this
 * rank's four partial residual sums), the sums are added in rank order on every rank (identical decisions, no
 * all-reduce), tolerances use nvar_global.  exaadmm_objective / info.objval stay this partition's share.
 * The exchange runs on a second stream BESIDE the next iteration's interior branch solves (only a cut branch's next solve
 * needs the halo); a batch of iterations including its exchanges is replayed from one hipGraph when the bound RCCL can
 * be captured (probed at connect time), enqueued directly otherwise.
 *   exaadmm_dist_unique_id  on ONE rank; the 128 bytes reach the others by the caller's means (torch.distributed, MPI)
 *   exaadmm_dist_prepare    local: binds RCCL, checks the cut lists, allocates the message buffers.  ncut_per_peer
 *                           [nranks]; cut_branch: for every peer in rank order the zero-based local ids of the
 *                           branches shared with it, in the order both sides agree on (ascending global branch id)
 *   exaadmm_dist_connect    collective: ncclCommInitRank.  Callers agree on the outcome of prepare across the ranks
 *                           first (a rank that failed alone would leave the others waiting here forever)
 *   exaadmm_dist_attach     prepare + connect in one call (nranks = 1: no id needed, nothing is exchanged) */
int exaadmm_dist_unique_id(void *out128, int64_t nbytes);
int exaadmm_dist_prepare(exaadmm_handle *h, int nranks, int rank, int64_t nvar_global, const int64_t *ncut_per_peer,
                         const int64_t *cut_branch);
int exaadmm_dist_connect(exaadmm_handle *h, const void *unique_id, int64_t id_bytes);
int exaadmm_dist_attach(exaadmm_handle *h, int nranks, int rank, const void *unique_id, int64_t id_bytes,
                        int64_t nvar_global, const int64_t *ncut_per_peer, const int64_t *cut_branch);
int exaadmm_dist_detach(exaadmm_handle *h);
/* Resident loop of a partition with peers: the in-library loop above keeps ONE launch per inner loop (exaadmm_loop_kind = 2) when the
 * ranks can write into each other's device memory -- what crosses GPUs (per cut branch the update of the bus at its far end, 4 doubles
 * each way; the four residual sums) then goes by direct stores over xGMI into the peer's MAILBOX from inside the launch, no kernel
 * boundary and no collective call per iteration.
 *   exaadmm_dist_p2p_export  after exaadmm_dist_prepare (and after the options / stream of the handle are final): 256 bytes describing this
 *                            rank's mailbox (process, device, address, IPC handle) and WHICH resident form this partition would run (none,
 *                            one tile, two tiles per wavefront -- size against the chip, options "resident" / "rec" / "spec", own stream)
 *   exaadmm_dist_p2p_import  every rank's blob, in rank order (the caller's all-gather): maps the peers' mailboxes.  Refused -- on every rank
 *                            alike, they read the same blobs -- when ANY rank's partition would not run a resident form: a rank on the
 *                            two-launch loop would leave its peers' launches in their capped waits.  Local otherwise; the ranks agree on the
 *                            outcome before they rely on it (a rank that cannot map a peer keeps two launches per iteration, and so must the
 *                            others: option "dist_p2p" = 0).  The first launch of an inner loop is split for the heavy wavefronts (option
 *                            "heavy_split") when every rank runs the one-tile form -- read from the blobs as well, so launch lengths agree.
 * The ranks wait for each other IN FRONT OF every resident launch and in every sums exchange between launches (||z|| of the outer loop, the
 * sums that complete a loop's last iteration), patiently: a peer may be up to 60 s behind.  Inside a launch the waits are short (~ 1 s).  A
 * launch whose waits run into their cap on any rank is reported to every rank by the launch's last kernel; all then restore the
 * iterate the call started from and repeat it with two launches per iteration + the RCCL exchange (exaadmm_loop_downgrades).  A sums exchange
 * that finds no peer within its 60 s fails the call (EXAADMM_ERR_HIP; exaadmm_last_error names the rank) -- it never returns a number. */
#define EXAADMM_P2P_BLOB_BYTES 256
int exaadmm_dist_p2p_export(exaadmm_handle *h, void *blob, int64_t nbytes);
int exaadmm_dist_p2p_import(exaadmm_handle *h, const void *blobs, int64_t nbytes_each);
/* the sums exchange + a ring send / receive of n doubles on the handle's stream; max_err = deviation from the
 * expected values.  Checks that RCCL and the library share one HIP runtime. */
int exaadmm_dist_selftest(exaadmm_handle *h, int64_t n, double *max_err);
/* the same ring send / receive captured into a hipGraph and replayed twice on the exchange stream: *captured = 1 when the bound
 * RCCL could be captured and delivered the right data from the replays, 0 when it refused capture (data then moved directly) */
int exaadmm_dist_capture_selftest(exaadmm_handle *h, int64_t n, int *captured);
/* what the distributed loop of this handle did so far: ranks of the communicator as RCCL counts them (ncclCommCount; -1 if the
 * bound library does not say), RCCL version code (ncclGetVersion; 0 unknown), mean duration in microseconds of the sampled
 * exchanges (grouped send / receive + unpack kernel, HIP events on the exchange stream; 0: none sampled), batches of
 * iterations replayed from a hipGraph / enqueued directly, and the result of the connect-time capture probe (1 = the bound
 * RCCL captured, instantiated and replayed an exchange with every peer, 0 = it refused: batches are enqueued directly) */
int exaadmm_dist_stats(exaadmm_handle *h, int *comm_ranks, int *rccl_version, double *exchange_us, int64_t *graph_batches,
                       int64_t *direct_batches, int *probe);
const char *exaadmm_dist_rccl_path(void); /* the RCCL the library bound (next to its own HIP runtime), "" before */
/* process-wide, before the first bind: the communication library to load instead of the librccl next to the HIP runtime
 * (a site's own build; the tests' in-process stand-in).  NULL / "" restores the default search. */
int exaadmm_dist_set_rccl_library(const char *path);
int64_t exaadmm_dist_exchange_count(void); /* grouped exchanges enqueued by the calling thread so far (tests) */

/* Caller-driven variant (torch.distributed / gloo / debugging): the step-wise entry points plus these helpers. */
/* run on a caller-provided HIP stream (e.g. torch's current stream).  NULL is the device's
 * default (null) stream -- which is what torch.cuda.current_stream().cuda_stream returns unless a
 * side stream is active; EXAADMM_OWN_STREAM switches back to the handle's private stream. */
#define EXAADMM_OWN_STREAM ((void *)(~(uintptr_t)0))
int exaadmm_set_stream(exaadmm_handle *h, void *hip_stream);
/* async != 0: step-wise entry points only enqueue (no stream synchronisation on return);
 * the caller orders them against its collectives through the stream */
int exaadmm_set_async(exaadmm_handle *h, int async);
/* halo lists: entry i moves element idx[i] (zero-based reference-layout index) of solution
 * vector which[i] (exaadmm_vector).  pack gathers into a contiguous DEVICE buffer, unpack
 * scatters from one; 8 slots per handle */
int exaadmm_set_halo_index(exaadmm_handle *h, int slot, const int32_t *which, const int64_t *idx, int64_t count);
int exaadmm_halo_pack(exaadmm_handle *h, int slot, double *dev_buf);
int exaadmm_halo_unpack(exaadmm_handle *h, int slot, const double *dev_buf);
/* fused z_prev/z/l update of every local element; the 4 partial sums of squares
 * {|rp|^2, |rd|^2, |z|^2, |Ax+By|^2} over the OWNED elements are written to dev_out4 (DEVICE memory) for an
 * all-reduce by the caller.  exaadmm_sumsq_z_partial: dev_out4[2] = sum z^2 of the owned elements. */
int exaadmm_update_zl_partial(exaadmm_handle *h, double beta, double *dev_out4);
int exaadmm_sumsq_z_partial(exaadmm_handle *h, double *dev_out4);
/* Device-resident loop control for a caller-driven distributed inner loop (same Ctrl block the fused
 * loop uses).  ctrl_begin arms it for outer iteration `outer`; while armed, the
 * step-wise entry points become no-ops on the device once the loop has converged, the branch
 * kernel takes major_iter = completed iterations + 1, z/l updates take beta from the block.
 * ctrl_step consumes the ALL-REDUCED sums (device pointer): inner++, norms = sqrt(sums),
 * done = primres <= sqrt(nvar_global)/(2500*outer) or inner == max_inner.  ctrl_read syncs. */
int exaadmm_ctrl_begin(exaadmm_handle *h, int outer, int max_inner, double beta, int64_t nvar_global);
int exaadmm_ctrl_step(exaadmm_handle *h, const double *dev_sums4);
int exaadmm_ctrl_read(exaadmm_handle *h, int *done, int *inner, double res[4]);
int exaadmm_ctrl_end(exaadmm_handle *h);

/* ---- multi-period ACOPF: ModelMpacopf (src/models/mpacopf/mpacopf_model.jl:40-112) ----
 * T periods of the same network with their own loads, coupled through generator ramp constraints
 *   p_{t,g} - phat_{t-1,g} - s_{t,g} = 0,  |s| <= ramp_ratio * pgmax_g,  phat_{t-1,g} in consensus with pbar_{t-1,g}.
 * The handle is ONE time-expanded model: every vector of the single-period API has T x (its
 * single-period length) entries in the layout [gens of period 1 | ... | gens of period T | lines of
 * period 1 | ... | lines of period T] (exaadmm_nvar = T * (2 ngen + 8 nline)), and every entry point
 * above applies to all periods at once with the reference's multi-period semantics:
 *   exaadmm_update_x      mpacopf_admm_update_x_cpu.jl:5-45 (closed form for period 1, 3-variable
 *                         augmented-Lagrangian x TRON ramp subproblem per generator for t >= 2, then lines)
 *   exaadmm_update_xbar   mpacopf_admm_update_xbar_cpu.jl / bus_kernel_ramp (mpacopf_bus_kernel_cpu.jl:1-121)
 *   exaadmm_update_z/l/lz mpacopf_admm_update_{z,l,lz}_cpu.jl (ramp block included)
 *   exaadmm_update_residual, exaadmm_norm_z   per-period norms incl. ramp block, MAX over periods
 *                         (mpacopf_admm_update_residual_cpu.jl:1-55, mpacopf_admm_prepoststep_cpu.jl:1-22)
 *   exaadmm_run_inner, exaadmm_solve_two_level  eps_pri / outer tolerance from mod.nvar = 2 ngen + 8 nline + ngen
 *   exaadmm_objective     sum over periods (mpacopf_admm_prepoststep_cpu.jl:73)
 * Pd, Qd: T x nbus, period-major (the columns of get_load, src/utils/opfdata.jl:121-130, one after another).
 * exaadmm_update_x_gen is replaced by exaadmm_mp_update_x_gen (the ramp subproblem needs major_iter);
 * the partition helpers (halo, masks, ctrl_*) are not available on a multi-period handle. */
int exaadmm_mp_create(const exaadmm_grid *g, const exaadmm_params *p, int T, const double *Pd, const double *Qd,
                      double ramp_ratio, int device_id, exaadmm_handle **out);
int exaadmm_mp_periods(exaadmm_handle *h);
int64_t exaadmm_mp_nvar(exaadmm_handle *h); /* mod.nvar of the reference */
int exaadmm_mp_update_x_gen(exaadmm_handle *h, int major_iter); /* mpacopf_admm_update_x_cpu.jl:5-31 */
/* SolutionRamping vectors (mpacopf_model.jl:1-38): T x ngen, period-major; block 0 (period 1) is unused.
 * `which`: U_CURR (phat), L_CURR, RHO, RD, RP, Z_CURR, Z_PREV, LZ, AX_PLUS_BY, RAMP_S, RAMP_MU, RAMP_XI */
int exaadmm_mp_download_ramp(exaadmm_handle *h, int which, double *host, int64_t n);
int exaadmm_mp_upload_ramp(exaadmm_handle *h, int which, const double *host, int64_t n);

/* ---- MATPOWER ingest (next row f1: utils/parse_matpower.jl, utils/opfdata.jl) ---- */
int exaadmm_load_matpower(const char *path, double tight_factor, const char *case_format, exaadmm_case **out);
int exaadmm_case_grid(const exaadmm_case *c, exaadmm_grid *out); /* zero-based view, valid until free */
int exaadmm_case_free(exaadmm_case *c);

#ifdef __cplusplus
}
#endif
#endif
