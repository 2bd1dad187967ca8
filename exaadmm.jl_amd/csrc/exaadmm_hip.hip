// libexaadmm_hip.so -- host side of the C ABI declared in include/exaadmm_hip.h.
// Handle management, reference-layout <-> SoA conversion, one entry point per plugin step
// of the reference (docs/src/dev.md:65-155) and the fused device-resident loops.
// There is NO CPU execution path in this library: without a gfx950 device every compute
// entry point fails with EXAADMM_ERR_NO_DEVICE.
#include "exaadmm_hip.h"
#include "kernels.h"

#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace exa;

static thread_local std::string g_err;

static void set_err(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return EXAADMM_ERR_HIP;                                                         \
        }                                                                                   \
    } while (0)

#define REQUIRE(cond, msg)            \
    do {                              \
        if (!(cond)) {                \
            set_err("%s", msg);       \
            return EXAADMM_ERR_INVALID; \
        }                             \
    } while (0)

static inline int round_up(int n, int m) { return ((n + m - 1) / m) * m; }

// host-side data needed only by init_solution! (model construction)
struct InitData {
    std::vector<double> Vmin, Vmax, pgmin, pgmax, qgmin, qgmax, Y[8];
    std::vector<int> brBus;
};

struct HaloSlot {
    int* d_which = nullptr;
    int* d_idx = nullptr;
    int count = 0;
};

struct exaadmm_dist;   // RCCL communicator + halo buffers of a partition (dist_native.h)
extern "C" int exaadmm_dist_detach(exaadmm_handle* h);

// A/B switches and diagnostics of a handle (exaadmm_set_option; include/exaadmm_hip.h lists them).  The library reads no
// environment variable of its own: a host that wants its switches from the environment translates them.
struct Options {
    int bus_kernel_lane = 0, graph = 1, spec = 1, rec = 1;
    int resident = 1, resident_tiles = 1, heavy = 1, heavy_max = 512, heavy_split = 48, heavy_split_prev = 96, heavy_adopt = 4, block_order = 1;
    int dist_overlap = 1, dist_p2p = 1;
    int trust_resident = 0;
    // -- not options of the product: switches of the test build (exaadmm_test_hook, -DEXA_TEST_HOOKS) and of the diagnostic build --
    int resident_coop = 1;        // 0: plain launch of the resident kernels (two ranks sharing ONE GPU in tests: the runtime refuses a second cooperative grid)
    int resident_cap_limit = 0;   // > 0: pretend the chip holds no more resident wavefronts than this (the two-tile form on small grids)
    int heavy_select_small = 0;   // 1: the kernel that picks the heavy wavefronts runs as ONE wavefront (a block of 16 cannot start on a GPU whose CUs all hold a
                                  // resident wavefront of ANOTHER launch that waits for this one -- ranks sharing a GPU in tests)
    int prof_resident = 0;        // diagnostic build (-DEXA_PROFILE_PHASES): phase clocks of the resident loop's lanes instead of k_branch_t's timelines
};
// tuning constants of the heavy wavefronts that used to be options (rounds 4-5 measured them: profiles/r05_honest_replays.txt)
constexpr int HEAVY_FROM = 64;     // first iteration of an inner loop from which a resident launch may carry heavy wavefronts
constexpr int HEAVY_WINDOW = 64;   // W: a branch that needed >= 3 solver trips in the last W (then 4 W, 16 W) resident iterations is a candidate

struct exaadmm_handle {
    exaadmm_dist* dist = nullptr;
    Options opt;
    int device = 0;
    int G = 0, L = 0, B = 0, Gp = 0, Lp = 0, N = 0;
    int64_t nvar = 0;
    // multi-period model (exaadmm_mp_create): T periods stored period-major in one time-expanded
    // handle; nvar_ctrl is mod.nvar of the reference (one period + ngen, mpacopf_model.jl:99-104),
    // which sets eps_pri and the outer tolerance
    int T = 1, g1 = 0, l1 = 0, b1 = 0, mp_nbx = 1;
    int64_t nvar_ctrl = 0;
    double baseMVA = 100.0;
    exaadmm_params par;
    Dev d;
    BranchOptions bo;
    Ctrl* d_ctrl = nullptr;
    Dev* d_dev = nullptr;       // device copy of `d` for kernels that take it by pointer
    bool dev_dirty = true;
    double* d_sums = nullptr;  // 4 doubles
    Ctrl* h_ctrl = nullptr;    // pinned
    double* h_sums = nullptr;  // pinned, 4 doubles
    hipStream_t own_stream = nullptr, stream = nullptr;
    int ew_blocks = 1;
    std::vector<void*> allocs;
    std::vector<double> c2, c1, c0;  // host copies for the objective
    std::vector<double> stage;       // host staging, N doubles
    std::vector<double> rateA_h;
    // device position of reference branch b.  Branches are stored sorted by the bus they are
    // gathered from (from-bus for own branches, to-bus for a partition's replicas of a peer's cut branches) so that the bus kernel's
    // gathers of consecutive buses hit consecutive elements of the planes.
    std::vector<int> perm;
    // per reference branch: 0 = both ends on this handle's buses, 1 = only the from-end, 2 = only the to-end (a partition's
    // cut branches; the other end's bus update runs on a peer)
    std::vector<char> cut_side;
    std::vector<double*> snap;   // exaadmm_state_save: u, v, z, z_prev, l, lz (N each), ls1, ls2, mu (Lp each)
    // ... and what the choice of the heavy wavefronts / the pairing of the tiles is made from (the solver-trip counters of every buffer set, the
    // trip history, its clock): a restored state that kept the counters of whatever ran last would give a REPLAYED run a heavy set chosen with
    // knowledge of its own future (seen: the repeats of bench.py's region 14 % faster than its first pass)
    int* snap_hist[4] = {nullptr, nullptr, nullptr, nullptr};   // st_ticks of sets 0 .. 2, last3
    bool snap_hist_valid[4] = {false, false, false, false};
    int snap_resident_iters = 0, snap_last_inner_len = -1, snap_nsets = -1;
    double* d_busc = nullptr;   // rho-only constants of the bus update (k_bus_const); recomputed when rho changes
    bool busc_dirty = true;
    // hipGraph of one batch of fused inner iterations (exaadmm_run_inner); rebuilt when anything
    // baked into its kernel arguments changes (parameters, attach / detach of a partition)
    hipGraphExec_t batch_graph = nullptr;
    bool use_graph = true;
    HaloSlot halo[8];
    bool residual_vectors_valid = false;
    bool async_mode = false;
    bool bus_lane_kernel = false;
    // speculative loop (kernels.h, k_branch_t<MP, true>): second set of the buffers the branch kernel writes,
    // the two Dev variants (odd / even iteration) on host and device, a pinned word for the batch base.
    // EXAADMM_NO_SPEC=1 (A/B switch) runs the plain four-launch loop instead.
    bool spec_on = false;
    // the iterate rotates through `nsets` buffer sets (IterSet): launch i of a loop reads set (i - 1) mod nsets and writes set
    // i mod nsets; sets[nsets - 1] is the PRIMARY set (h->d's own buffers: where every loop starts and where spec_finish brings
    // the result back).  One GPU: 2 sets, z / l / z_prev updated in place.  A partition with peers: 3 sets, z and l rotate
    // as well (kernels.h Dev::z_in), z_prev is materialised at the end of the loop.
    struct IterSet {
        double *u = nullptr, *v = nullptr, *ls1 = nullptr, *ls2 = nullptr, *mu = nullptr, *z = nullptr, *l = nullptr;
        int* st[3] = {nullptr, nullptr, nullptr};
        double *r_u = nullptr, *r_s = nullptr, *r_mu = nullptr, *r_xi = nullptr;   // multi-period model: what the ramp subproblems write
    };
    IterSet sets[3], alt[2];
    int nsets = 0;
    double *zp_scratch = nullptr, *partials_br = nullptr;
    // branch blocks [0, nb_int) hold only branches whose two buses are local (exaadmm_create stores the cut branches last)
    int n_interior = 0;
    // resident loop (kernels.h k_resident): the default loop of a single-period grid without peers whose wavefronts all fit
    // the chip at once (EXAADMM_NO_RESIDENT=1: A/B switch).  resident_cap = wavefronts of k_resident this device holds
    // at once (computed once per handle); resident_downgrades = batches that ran into a wait cap (the GPU was shared):
    // the iterate was restored from `guard` and the batch repeated with the two-launch loop, which the handle then keeps.
    bool resident_ok = false;
    int resident_cap = -1;
    int resident_downgrades = 0;
    int resident_launches = 0;
    bool resident_zeroed = false;   // the hand-over words were reset by the call's opening launch (k_loop_begin)
    bool resident_coop = false;
    // what the last loop call launched (exaadmm_resident_form) and the first-use self checks (exaadmm_resident_selfcheck_stats)
    int form_tiles = 0, form_heavy = 0, form_launches = 0, form_grid = 0;
    bool checked_resident = false, checked_heavy = false, in_selfcheck = false;
    bool force_two_launch = false;   // self check: the comparison leg runs the two-launch loop on a handle that is resident-eligible
    int force_heavy = -1;            // self check: -1 the library's own rule, 0 / 1 the kernel form the checked call used
    unsigned form_used = 0;          // kernel forms the current call launched: bit 0 plain / two-tile, bit 1 heavy
    int selfchecks = 0, selfcheck_mismatches = 0;
    std::vector<void*> chk_e, chk_r; // self check: the iterate the checked call produced / what the resident leg made of it
    unsigned long long* d_chk_diff = nullptr;
    double* rec_alloc = nullptr;     // the pushed-records buffer (d.rec points at it unless option "rec" = 0)
#if defined(EXA_TEST_HOOKS)
    int hook_fail_launch = 0;        // test build only: the n-th resident launch of the handle reports a failed wait
    int hook_flip_ulp = 0;           // test build only: the next resident launch's result gets one ulp added to one u entry
    int hook_p2p_delay_ticks = 0;    // test build only: every mailbox send of the resident loop with peers leaves this many 100 MHz ticks late
#endif
    int resident_cap2 = -1;      // the same for k_resident_t<2> (two tiles of 64 branches per wavefront)
    // resident loop of a partition WITH PEERS (kernels.h ResidentPeers; dist_native.h "mailboxes"): what crosses GPUs goes by direct
    // stores into the peer's mailbox.  p2p_ready: every peer's mailbox is mapped (exaadmm_dist_p2p_import); resident_peers_ok: the
    // grid's bus tables fit the loop's counters
    bool resident_peers_ok = false, p2p_ready = false;
    bool peers_one_tile = false;   // every rank of the partition runs the one-tile resident form (exaadmm_dist_p2p_import): launch lengths may follow heavy_split
    ResidentPeers rpeers = {};
    int p2p_epoch = 0;
    ResidentTiles rtiles = {nullptr, 0, 0, 0, 0};
    ResidentHeavy rheavy = {nullptr, nullptr, nullptr, nullptr, 0};   // heavy wavefronts of the resident launch (kernels.h); n = 0: none
    int heavy_max = 0;
    bool blk_order_off = false;
    int *blk_order = nullptr, *blk_keys = nullptr;   // Dev::blk_order of the whole-grid launches of the fused loops (grids with more blocks than SIMDs)
    int last_inner_len = -1;  // iterations of the handle's previous convergence-tested inner loop (run_inner_spec); -1: there was none yet
    int resident_iters = 0;   // iterations the resident launches of this handle were asked for so far (the clock of ResidentSync::last3)
    std::vector<void*> guard;
    size_t nrec_slots = 0;
    ResidentSync rsync = {};
    // Dev variants of the fused loops, [i mod nsets][0 = whole launch, 1 = interior blocks + generators, 2 = cut blocks], on
    // host and device; dfin[r]: what the pass that completes a loop's last iteration sees (spec_complete_last)
    Dev dspec[3][3];
    Dev* d_dspec[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    Dev dfin[3];
    int* h_base = nullptr;
    hipGraphExec_t spec_graph = nullptr;
    bool ctrl_mode = false;
    exaadmm_info info;
    std::vector<hipEvent_t> ev;
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};  // sampled per-step timing of the fused loop
    double acc_ms[3] = {0.0, 0.0, 0.0};                          // x-update, xbar-update, z/l/residual
    InitData init;
};


static int attach_init_data(exaadmm_handle* h, const exaadmm_grid* g);
static void launch_gen(exaadmm_handle* h, const Ctrl* c, int major_iter = 0);
static int sync_check(exaadmm_handle* h, bool force = false);
template <int MODE>
static void launch_mp_elem(exaadmm_handle* h, double beta, const Ctrl* c);

template <class T>
static int dev_alloc(exaadmm_handle* h, T** p, size_t n)
{
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return 0;
}

// reference index (0-based) -> SoA index
static inline int soa_index(const exaadmm_handle* h, int64_t r)
{
    if (r < 2 * (int64_t)h->G) return (int)((r & 1) * h->Gp + (r >> 1));
    const int64_t q = r - 2 * (int64_t)h->G;
    return (int)(2 * h->Gp + (q & 7) * h->Lp + h->perm[q >> 3]);
}

static void ref_to_soa(const exaadmm_handle* h, const double* ref, double* soa, double pad)
{
    for (int i = 0; i < h->N; i++) soa[i] = pad;
    for (int i = 0; i < h->G; i++) {
        soa[i] = ref[2 * i];
        soa[h->Gp + i] = ref[2 * i + 1];
    }
    const double* rb = ref + 2 * (size_t)h->G;
    for (int k = 0; k < 8; k++) {
        double* plane = soa + 2 * h->Gp + (size_t)k * h->Lp;
        for (int b = 0; b < h->L; b++) plane[h->perm[b]] = rb[8 * (size_t)b + k];
    }
}

static void soa_to_ref(const exaadmm_handle* h, const double* soa, double* ref)
{
    for (int i = 0; i < h->G; i++) {
        ref[2 * i] = soa[i];
        ref[2 * i + 1] = soa[h->Gp + i];
    }
    double* rb = ref + 2 * (size_t)h->G;
    for (int k = 0; k < 8; k++) {
        const double* plane = soa + 2 * h->Gp + (size_t)k * h->Lp;
        for (int b = 0; b < h->L; b++) rb[8 * (size_t)b + k] = plane[h->perm[b]];
    }
}

static double* vec_ptr(exaadmm_handle* h, int which)
{
    switch (which) {
        case EXAADMM_U_CURR: return h->d.u;
        case EXAADMM_V_CURR: return h->d.v;
        case EXAADMM_L_CURR: return h->d.l;
        case EXAADMM_RHO: return h->d.rho;
        case EXAADMM_RD: return h->d.rd;
        case EXAADMM_RP: return h->d.rp;
        case EXAADMM_Z_CURR: return h->d.z;
        case EXAADMM_Z_PREV: return h->d.zp;
        case EXAADMM_LZ: return h->d.lz;
        case EXAADMM_AX_PLUS_BY: return h->d.ax;
        default: return nullptr;
    }
}

#ifndef EXA_BUILD_FLAGS
#error "build through csrc/Makefile: it pins the code-generation flags the branch kernel needs (EXA_BUILD_FLAGS)"
#endif
// the flags the library was compiled with travel in the version string: k_branch is only known to be
// compiled correctly with -mllvm -sgpr-regalloc=basic (DESIGN.md "compiler note"); loaders check for it
#ifndef EXA_SRC_HASH
#define EXA_SRC_HASH "unknown"
#endif
extern "C" const char* exaadmm_version(void) { return "exaadmm-hip 0.5 (gfx950) [" EXA_BUILD_FLAGS "] src:" EXA_SRC_HASH; }
extern "C" const char* exaadmm_last_error(void) { return g_err.c_str(); }

// Number of visible HIP devices (0 when there is none or the runtime cannot start).  Also the library's way of
// starting ITS HIP runtime at a defined moment: a process that holds a second HIP runtime (PyTorch bundles one)
// must not start that one between this library's load and its first HIP call (observed: this runtime then reports
// no device); the Python loader therefore calls this right after loading the library.
extern "C" int exaadmm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// Which HIP runtime the library's hip* symbols are bound to in this process (a Python process that has PyTorch maps
// torch's bundled runtime first, a C / Julia host gets the system one), its version, and the version of the headers the
// library was compiled against.  Returns the runtime's major version minus the headers' (0 = compatible).
extern "C" int exaadmm_hip_runtime_info(char* out, int64_t n)
{
    Dl_info info;
    const char* path = "?";
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) path = info.dli_fname;
    int rv = 0;
    if (hipRuntimeGetVersion(&rv) != hipSuccess) {
        (void)hipGetLastError();
        rv = 0;
    }
    if (out && n > 0)
        snprintf(out, (size_t)n, "%s (runtime version %d.%d.%d; library compiled against HIP %d.%d.%d)", path, rv / 10000000,
                 (rv / 100000) % 100, rv % 100000, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH);
    return rv / 10000000 - HIP_VERSION_MAJOR;
}

extern "C" void exaadmm_default_params(exaadmm_params* p)
{
    // src/utils/environment.jl:43-75
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
    p->verbose = 0;
    p->check_every = 1;
}

static void invalidate_graph(exaadmm_handle* h)
{
    if (h->batch_graph) {
        (void)hipGraphExecDestroy(h->batch_graph);
        h->batch_graph = nullptr;
    }
    if (h->spec_graph) {
        (void)hipGraphExecDestroy(h->spec_graph);
        h->spec_graph = nullptr;
    }
}

static void fill_branch_options(exaadmm_handle* h)
{
    invalidate_graph(h);
    BranchOptions& o = h->bo;
    o.max_auglag = h->par.max_auglag;
    o.mu_max = h->par.mu_max;
}

static int upload_mupow_table()
{
    MuPowTable t;
    double p = 1.0;
    for (int k = 0; k < 8; k++) {
        p *= 10.0;
        t.pow10[k] = p;
        t.pow01[k] = pow(p, 0.1);
        t.pow09[k] = pow(p, 0.9);
    }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_mupow), &t, sizeof(t)));
    return 0;
}

template <class T>
static int upload_vec(exaadmm_handle* h, T* dst, const std::vector<T>& src)
{
    if (!src.empty()) HIPCHK(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int build_csr(exaadmm_handle* h, const int64_t* start, const int64_t* idx, int base, int nbus, int64_t nnz_expected,
                     const int** d_start, const int** d_idx, int idx_limit, const std::vector<int>* map = nullptr)
{
    std::vector<int> s(nbus + 1), ix((size_t)std::max<int64_t>(nnz_expected, 0));
    for (int i = 0; i <= nbus; i++) s[i] = (int)(start[i] - base);
    // a partition lists only the ends that touch its own buses: nnz may be smaller than the count
    if (s[0] != 0 || s[nbus] < 0 || s[nbus] > nnz_expected) {
        set_err("CSR start array inconsistent (first=%d last=%d max nnz=%lld)", s[0], s[nbus], (long long)nnz_expected);
        return EXAADMM_ERR_INVALID;
    }
    for (int i = 0; i < nbus; i++)
        if (s[i + 1] < s[i]) {
            set_err("CSR start array not monotone");
            return EXAADMM_ERR_INVALID;
        }
    nnz_expected = s[nbus];
    ix.resize((size_t)nnz_expected);
    for (int64_t k = 0; k < nnz_expected; k++) {
        ix[k] = (int)(idx[k] - base);
        if (ix[k] < 0 || ix[k] >= idx_limit) {
            set_err("CSR index out of range");
            return EXAADMM_ERR_INVALID;
        }
        if (map) ix[k] = (*map)[ix[k]];
    }
    int *ds = nullptr, *di = nullptr;
    int rc;
    if ((rc = dev_alloc(h, &ds, s.size()))) return rc;
    if ((rc = dev_alloc(h, &di, ix.size()))) return rc;
    if ((rc = upload_vec(h, ds, s))) return rc;
    if ((rc = upload_vec(h, di, ix))) return rc;
    *d_start = ds;
    *d_idx = di;
    return 0;
}

extern "C" int exaadmm_create(const exaadmm_grid* g, const exaadmm_params* p, int device_id, exaadmm_handle** out)
{
    REQUIRE(g && out, "null argument");
    REQUIRE(g->ngen >= 0 && g->nline > 0 && g->nbus > 0, "empty grid");
    REQUIRE(g->index_base == 0 || g->index_base == 1, "index_base must be 0 or 1");
    REQUIRE(2 * g->ngen + 8 * g->nline < (int64_t)2000000000, "grid too large for 32-bit device indices");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_err("no HIP device visible: this backend has no CPU fallback");
        return EXAADMM_ERR_NO_DEVICE;
    }
    REQUIRE(device_id >= 0 && device_id < ndev, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err("device %d is %s; this library carries gfx950 (MI355X) code only", device_id, prop.gcnArchName);
        return EXAADMM_ERR_NO_DEVICE;
    }

    REQUIRE(g->nline_owned >= 0 && g->nline_owned <= g->nline, "nline_owned out of range");
    exaadmm_handle* h = new exaadmm_handle();
    h->device = device_id;
    h->G = (int)g->ngen;
    h->L = (int)g->nline;
    h->B = (int)g->nbus;
    h->Gp = round_up(std::max(h->G, 1), 64);
    h->Lp = round_up(h->L, 64);
    h->N = 2 * h->Gp + 8 * h->Lp;
    h->nvar = 2 * (int64_t)h->G + 8 * (int64_t)h->L;
    h->nvar_ctrl = h->nvar;
    h->g1 = h->G; h->l1 = h->L; h->b1 = h->B;
    h->baseMVA = g->baseMVA;
    if (p)
        h->par = *p;
    else
        exaadmm_default_params(&h->par);
    fill_branch_options(h);
    {
        // storage order of the branches: own branches sorted by from-bus, a partition's replicas of a peer's cut branches (to-bus
        // local) after them, sorted by to-bus -- and, among the own ones, those whose to-bus belongs to a peer LAST: every block
        // in front of the first cut branch can start its next solve without waiting for the peers' halo (split launches)
        const int64_t la = g->nline_owned > 0 ? g->nline_owned : g->nline;
        std::vector<char> in_fr(h->L, 0), in_to(h->L, 0);
        // (the CSR arrays are validated in full further down, by build_csr; here: present, and no more than nline entries are looked at)
        if (!(g->FrStart && g->FrIdx && g->ToStart && g->ToIdx && g->GenStart && (g->ngen == 0 || g->GenIdx) && g->brBusIdx)) {
            set_err("null CSR / bus index array");
            delete h;
            return EXAADMM_ERR_INVALID;
        }
        const int64_t nfr = std::min<int64_t>(std::max<int64_t>(g->FrStart[h->B] - g->index_base, 0), h->L);
        const int64_t nto = std::min<int64_t>(std::max<int64_t>(g->ToStart[h->B] - g->index_base, 0), h->L);
        for (int64_t k = 0; k < nfr; k++) {
            const int64_t b = g->FrIdx[k] - g->index_base;
            if (b >= 0 && b < h->L) in_fr[(size_t)b] = 1;
        }
        for (int64_t k = 0; k < nto; k++) {
            const int64_t b = g->ToIdx[k] - g->index_base;
            if (b >= 0 && b < h->L) in_to[(size_t)b] = 1;
        }
        std::vector<int> order(h->L);
        for (int b = 0; b < h->L; b++) order[b] = b;
        std::vector<int64_t> key(h->L);
        h->n_interior = 0;
        for (int b = 0; b < h->L; b++) {
            const bool own = b < la;
            const bool cut = in_fr[b] != in_to[b];
            const int64_t bus = g->brBusIdx[2 * (size_t)b + (own ? 0 : 1)] - g->index_base;
            key[b] = ((int64_t)(own ? (cut ? 1 : 0) : 2) << 40) + bus;
            if (own && !cut) h->n_interior++;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
        h->perm.assign(h->L, 0);
        for (int pos = 0; pos < h->L; pos++) h->perm[order[pos]] = pos;
    }
    memset(&h->info, 0, sizeof(h->info));
    memset(&h->d, 0, sizeof(h->d));
    Dev& d = h->d;
    d.G = h->G; d.L = h->L; d.B = h->B; d.Gp = h->Gp; d.Lp = h->Lp; d.N = h->N;
    d.Lown = g->nline_owned > 0 ? (int)g->nline_owned : h->L;
    d.nranks = 1;
    d.T = 1; d.g1 = h->G; d.l1 = h->L; d.b1 = h->B;
    const int base = g->index_base;
    int rc = 0;
#define TRY(x) do { if ((rc = (x))) { exaadmm_destroy(h); return rc; } } while (0)
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_err("hipStreamCreate failed");
        delete h;
        return EXAADMM_ERR_HIP;
    }
    h->stream = h->own_stream;
    // (defaults of the options: lane-group bus kernel, hipGraph batches, speculative two-launch loop -- which needs the lane-group
    //  bus kernel: it carries the reduction block; exaadmm_set_option changes them)
    h->bus_lane_kernel = false;
    h->use_graph = true;
    h->spec_on = true;
    TRY(upload_mupow_table());

    double** svecs[] = {&d.u, &d.v, &d.z, &d.zp, &d.l, &d.lz, &d.rho, &d.rp, &d.rd, &d.ax};
    for (auto pp : svecs) {
        TRY(dev_alloc(h, pp, (size_t)h->N));
        if (hipMemset(*pp, 0, (size_t)h->N * sizeof(double)) != hipSuccess) { set_err("hipMemset failed"); exaadmm_destroy(h); return EXAADMM_ERR_HIP; }
    }
    h->stage.assign(h->N, 0.0);

    // branch constants -> planes
    {
        std::vector<double> Y((size_t)8 * h->Lp, 0.0), bnd((size_t)8 * h->Lp, 0.0), rate(h->Lp, 0.0);
        const double* Ysrc[8] = {g->YffR, g->YffI, g->YftR, g->YftI, g->YttR, g->YttI, g->YtfR, g->YtfI};
        const double* Bsrc[4] = {g->FrVmBound, g->ToVmBound, g->FrVaBound, g->ToVaBound};
        for (int k = 0; k < 8; k++)
            for (int b = 0; b < h->L; b++) Y[(size_t)k * h->Lp + h->perm[b]] = Ysrc[k][b];
        for (int k = 0; k < 4; k++)
            for (int b = 0; b < h->L; b++) {
                bnd[(size_t)(2 * k) * h->Lp + h->perm[b]] = Bsrc[k][2 * b];
                bnd[(size_t)(2 * k + 1) * h->Lp + h->perm[b]] = Bsrc[k][2 * b + 1];
            }
        for (int b = 0; b < h->L; b++) rate[h->perm[b]] = g->rateA[b];
        h->rateA_h.assign(g->rateA, g->rateA + h->L);
        double *dY, *dB, *dR;
        TRY(dev_alloc(h, &dY, Y.size()));
        TRY(dev_alloc(h, &dB, bnd.size()));
        TRY(dev_alloc(h, &dR, rate.size()));
        TRY(upload_vec(h, dY, Y));
        TRY(upload_vec(h, dB, bnd));
        TRY(upload_vec(h, dR, rate));
        d.Y = dY; d.bnd = dB; d.rateA = dR;
        TRY(dev_alloc(h, &d.ls1, (size_t)h->Lp));
        TRY(dev_alloc(h, &d.ls2, (size_t)h->Lp));
        TRY(dev_alloc(h, &d.mu, (size_t)h->Lp));
        std::vector<double> zeros(h->Lp, 0.0);
        TRY(upload_vec(h, d.ls1, zeros));
        TRY(upload_vec(h, d.ls2, zeros));
        TRY(upload_vec(h, d.mu, zeros));
        TRY(dev_alloc(h, &d.st_auglag, (size_t)h->Lp));
        TRY(dev_alloc(h, &d.st_minor, (size_t)h->Lp));
        TRY(dev_alloc(h, &d.st_ticks, (size_t)h->Lp));
        TRY(dev_alloc(h, &d.st_cyc, (size_t)13 * h->Lp));
        TRY(dev_alloc(h, &d.st_tl, (size_t)2 * h->Lp));
        if (hipMemset(d.st_tl, 0, (size_t)2 * h->Lp * sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(d.st_cyc, 0, (size_t)13 * h->Lp * sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(d.st_auglag, 0, h->Lp * sizeof(int)) != hipSuccess ||
            hipMemset(d.st_minor, 0, h->Lp * sizeof(int)) != hipSuccess ||
            hipMemset(d.st_ticks, 0, h->Lp * sizeof(int)) != hipSuccess) {
            set_err("hipMemset failed");
            exaadmm_destroy(h);
            return EXAADMM_ERR_HIP;
        }
    }
    // generator constants
    {
        const double* src[6] = {g->pgmin, g->pgmax, g->qgmin, g->qgmax, g->c2, g->c1};
        const double** dst[6] = {&d.pgmin, &d.pgmax, &d.qgmin, &d.qgmax, &d.c2, &d.c1};
        for (int k = 0; k < 6; k++) {
            std::vector<double> t(h->Gp, 0.0);
            for (int i = 0; i < h->G; i++) t[i] = src[k][i];
            double* dp;
            TRY(dev_alloc(h, &dp, t.size()));
            TRY(upload_vec(h, dp, t));
            *dst[k] = dp;
        }
        h->c2.assign(g->c2, g->c2 + h->G);
        h->c1.assign(g->c1, g->c1 + h->G);
        h->c0.assign(g->c0, g->c0 + h->G);
    }
    // bus constants + CSR
    {
        const double* src[4] = {g->Pd, g->Qd, g->YshR, g->YshI};
        const double** dst[4] = {&d.Pd, &d.Qd, &d.YshR, &d.YshI};
        for (int k = 0; k < 4; k++) {
            std::vector<double> t(src[k], src[k] + h->B);
            double* dp;
            TRY(dev_alloc(h, &dp, t.size()));
            TRY(upload_vec(h, dp, t));
            *dst[k] = dp;
        }
        TRY(build_csr(h, g->FrStart, g->FrIdx, base, h->B, h->L, &d.FrStart, &d.FrIdx, h->L, &h->perm));
        TRY(build_csr(h, g->ToStart, g->ToIdx, base, h->B, h->L, &d.ToStart, &d.ToIdx, h->L, &h->perm));
        TRY(build_csr(h, g->GenStart, g->GenIdx, base, h->B, h->G, &d.GenStart, &d.GenIdx, std::max(h->G, 1)));
        // combined per-bus component list for the lane-group bus kernel: generators, from-ends, to-ends
        {
            // a branch with only ONE end on a local bus is shared with a peer partition: the v entries of that end are what
            // the peer is missing (END_CUT -> the bus kernel also drops them into the halo message, dist_native.h)
            std::vector<char> seen_fr(h->L, 0), seen_to(h->L, 0);
            for (int64_t k = 0; k < g->FrStart[h->B] - base; k++) seen_fr[(size_t)(g->FrIdx[k] - base)] = 1;
            for (int64_t k = 0; k < g->ToStart[h->B] - base; k++) seen_to[(size_t)(g->ToIdx[k] - base)] = 1;
            h->cut_side.assign(h->L, 0);
            for (int b = 0; b < h->L; b++) h->cut_side[b] = (seen_fr[b] && !seen_to[b]) ? 1 : ((!seen_fr[b] && seen_to[b]) ? 2 : 0);
            std::vector<int> es(h->B + 1, 0), ec;
            ec.reserve((size_t)2 * h->L + h->G);
            for (int I = 0; I < h->B; I++) {
                es[I] = (int)ec.size();
                for (int64_t k = g->GenStart[I] - base; k < g->GenStart[I + 1] - base; k++)
                    ec.push_back((END_GEN << 28) | (int)(g->GenIdx[k] - base));
                for (int64_t k = g->FrStart[I] - base; k < g->FrStart[I + 1] - base; k++) {
                    const size_t b = (size_t)(g->FrIdx[k] - base);
                    ec.push_back((END_FROM << 28) | h->perm[b] | (seen_to[b] ? 0 : END_CUT));
                }
                for (int64_t k = g->ToStart[I] - base; k < g->ToStart[I + 1] - base; k++) {
                    const size_t b = (size_t)(g->ToIdx[k] - base);
                    ec.push_back((END_TO << 28) | h->perm[b] | (seen_fr[b] ? 0 : END_CUT));
                }
            }
            es[h->B] = (int)ec.size();
            int *des, *dec;
            TRY(dev_alloc(h, &des, es.size()));
            TRY(dev_alloc(h, &dec, ec.size()));
            TRY(upload_vec(h, des, es));
            TRY(upload_vec(h, dec, ec));
            d.EndStart = des;
            d.EndCode = dec;
            // slot layout of the first chunk of every bus (k_bus_group_t)
            std::vector<int> slot((size_t)h->B * BUS_W, 0), cnt(h->B, 0);
            for (int I = 0; I < h->B; I++) {
                cnt[I] = es[I + 1] - es[I];
                for (int k = 0; k < std::min(cnt[I], BUS_W); k++) slot[(size_t)I * BUS_W + k] = ec[(size_t)es[I] + k];
            }
            int *dsl, *dcn;
            TRY(dev_alloc(h, &dsl, slot.size()));
            TRY(dev_alloc(h, &dcn, cnt.size()));
            TRY(upload_vec(h, dsl, slot));
            TRY(upload_vec(h, dcn, cnt));
            d.SlotCode = dsl;
            d.EndCount = dcn;
            // "push" records of the fused loop (kernels.h, Dev::rec): where every generator / branch end drops what the bus
            // update needs from it.  Component k of bus I: slot 8 I + k, from the 9th component on 8 B + (position in EndCode)
            {   // (option "rec" = 0, A/B switch: the fused loop then gathers like the step-wise path)
                std::vector<int> rbr(2 * (size_t)h->Lp, -1), rgen(h->Gp, -1);
                for (int I = 0; I < h->B; I++)
                    for (int k = 0; k < cnt[I]; k++) {
                        const int code = ec[(size_t)es[I] + k], kind = (code >> 28) & 3, c = code & END_INDEX;
                        const int rs = (k < BUS_W) ? I * BUS_W + k : BUS_W * h->B + es[I] + k;
                        if (kind == END_GEN)
                            rgen[c] = rs;
                        else
                            rbr[2 * (size_t)c + (kind == END_FROM ? 0 : 1)] = rs;
                    }
                int *drb, *drg;
                double* drec;
                const size_t nrec = ((size_t)BUS_W * h->B + ec.size()) * 8;
                TRY(dev_alloc(h, &drb, rbr.size()));
                TRY(dev_alloc(h, &drg, rgen.size()));
                TRY(dev_alloc(h, &drec, nrec));
                TRY(upload_vec(h, drb, rbr));
                TRY(upload_vec(h, drg, rgen));
                if (hipMemset(drec, 0, nrec * sizeof(double)) != hipSuccess) { set_err("hipMemset failed"); exaadmm_destroy(h); return EXAADMM_ERR_HIP; }
                d.rec_br = drb;
                d.rec_gen = drg;
                d.rec = drec;
                h->rec_alloc = drec;
                // resident loop: per branch (SoA index) its two buses and their counts, per generator its bus
                std::vector<int> bb4(4 * (size_t)h->Lp, 0), gb2(2 * (size_t)h->Gp, 0);
                bool all_local = true;
                for (int I = 0; I < h->B; I++) {
                    int ngen = 0;
                    for (int k = 0; k < cnt[I]; k++) ngen += (((ec[(size_t)es[I] + k] >> 28) & 3) == END_GEN);
                    const int packed = (ngen << 16) | cnt[I];
                    if (cnt[I] > 0xffff) all_local = false;
                    for (int k = 0; k < cnt[I]; k++) {
                        const int code = ec[(size_t)es[I] + k], kind = (code >> 28) & 3, c = code & END_INDEX;
                        if (kind == END_GEN) {
                            gb2[2 * (size_t)c] = I;
                            gb2[2 * (size_t)c + 1] = packed;
                        } else {
                            bb4[4 * (size_t)c + (kind == END_FROM ? 0 : 2)] = I;
                            bb4[4 * (size_t)c + (kind == END_FROM ? 1 : 3)] = packed;
                        }
                    }
                }
                for (int b = 0; b < h->L; b++) all_local = all_local && h->cut_side[b] == 0;
                int *dbb, *dgb;
                TRY(dev_alloc(h, &dbb, bb4.size()));
                TRY(dev_alloc(h, &dgb, gb2.size()));
                TRY(upload_vec(h, dbb, bb4));
                TRY(upload_vec(h, dgb, gb2));
                d.br_bus4 = dbb;
                d.gen_bus2 = dgb;
                h->nrec_slots = (size_t)BUS_W * h->B + ec.size();
                h->resident_ok = all_local && h->T == 1;
                bool counts_ok = true;
                for (int I = 0; I < h->B; I++) counts_ok = counts_ok && cnt[I] <= 0xffff;
                h->resident_peers_ok = counts_ok && h->T == 1;
            }
        }
    }
    // reduction scratch + loop control
    h->ew_blocks = std::min(2048, std::max(1, (h->N + EW_BLOCK - 1) / EW_BLOCK));
    TRY(dev_alloc(h, &d.partials, (size_t)h->ew_blocks * 4));
    TRY(dev_alloc(h, &h->d_ctrl, 1));
    TRY(dev_alloc(h, &h->d_dev, 1));
    TRY(dev_alloc(h, &h->d_sums, 4));
    if (hipHostMalloc((void**)&h->h_ctrl, sizeof(Ctrl), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_sums, 4 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        set_err("hipHostMalloc failed");
        exaadmm_destroy(h);
        return EXAADMM_ERR_HIP;
    }
    memset(h->h_ctrl, 0, sizeof(Ctrl));
    TRY(dev_alloc(h, &h->d_busc, (size_t)BUSC_N * h->B));
    d.busc = h->d_busc;
    TRY(attach_init_data(h, g));
#undef TRY
    *out = h;
    return EXAADMM_OK;
}

extern "C" int exaadmm_destroy(exaadmm_handle* h)
{
    if (!h) return EXAADMM_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    (void)exaadmm_dist_detach(h);
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto& s : h->halo) {
        if (s.d_idx) (void)hipFree(s.d_idx);
        if (s.d_which) (void)hipFree(s.d_which);
    }
    if (h->batch_graph) (void)hipGraphExecDestroy(h->batch_graph);
    if (h->spec_graph) (void)hipGraphExecDestroy(h->spec_graph);
    if (h->h_base) (void)hipHostFree(h->h_base);
    if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
    if (h->h_sums) (void)hipHostFree(h->h_sums);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    for (auto e : h->tev)
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return EXAADMM_OK;
}

extern "C" int exaadmm_set_params(exaadmm_handle* h, const exaadmm_params* p)
{
    REQUIRE(h && p, "null argument");
    h->par = *p;
    fill_branch_options(h);
    return EXAADMM_OK;
}

extern "C" int exaadmm_get_params(exaadmm_handle* h, exaadmm_params* p)
{
    REQUIRE(h && p, "null argument");
    *p = h->par;
    return EXAADMM_OK;
}

extern "C" int64_t exaadmm_nvar(exaadmm_handle* h) { return h ? h->nvar : -1; }

extern "C" int exaadmm_set_stream(exaadmm_handle* h, void* hip_stream)
{
    REQUIRE(h, "null handle");
    // NULL is a real stream (the null stream, torch's default); the sentinel selects the private one
    h->stream = (hip_stream == EXAADMM_OWN_STREAM) ? h->own_stream : (hipStream_t)hip_stream;
    return EXAADMM_OK;
}

// ------------------------------------------------------------------ host <-> device
extern "C" int exaadmm_upload_solution(exaadmm_handle* h, int which, const double* host, int64_t n)
{
    REQUIRE(h && host, "null argument");
    REQUIRE(n == h->nvar, "vector length must be nvar = 2*ngen + 8*nline");
    double* dst = vec_ptr(h, which);
    REQUIRE(dst, "unknown vector id");
    HIPCHK(hipSetDevice(h->device));
    ref_to_soa(h, host, h->stage.data(), which == EXAADMM_RHO ? 1.0 : 0.0);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(dst, h->stage.data(), (size_t)h->N * sizeof(double), hipMemcpyHostToDevice));
    if (which == EXAADMM_RHO) h->busc_dirty = true;
    return EXAADMM_OK;
}

static int materialize_if_needed(exaadmm_handle* h)
{
    if (!h->residual_vectors_valid) {
        if (h->T > 1)
            launch_mp_elem<MP_RESIDUAL>(h, 0.0, nullptr);   // also fills the ramp block's rp, rd, Ax_plus_By
        else
            hipLaunchKernelGGL(k_materialize, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d);
        HIPCHK(hipGetLastError());
        h->residual_vectors_valid = true;
    }
    return 0;
}

extern "C" int exaadmm_download_solution(exaadmm_handle* h, int which, double* host, int64_t n)
{
    REQUIRE(h && host, "null argument");
    REQUIRE(n == h->nvar, "vector length must be nvar = 2*ngen + 8*nline");
    double* src = vec_ptr(h, which);
    REQUIRE(src, "unknown vector id");
    HIPCHK(hipSetDevice(h->device));
    if (which == EXAADMM_RP || which == EXAADMM_RD || which == EXAADMM_AX_PLUS_BY) {
        int rc = materialize_if_needed(h);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->stage.data(), src, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost));
    soa_to_ref(h, h->stage.data(), host);
    return EXAADMM_OK;
}

extern "C" int exaadmm_upload_membuf(exaadmm_handle* h, const double* host, int64_t nrow, int64_t nline)
{
    REQUIRE(h && host, "null argument");
    REQUIRE(nrow == 31 && nline == h->L, "membuf must be 31 x nline");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> a(h->Lp, 0.0), b(h->Lp, 0.0), c(h->Lp, 0.0), r(h->Lp, 0.0);
    for (int i = 0; i < h->L; i++) {
        const int q = h->perm[i];
        a[q] = host[24 + 31 * (size_t)i];
        b[q] = host[25 + 31 * (size_t)i];
        c[q] = host[26 + 31 * (size_t)i];
        r[q] = host[28 + 31 * (size_t)i];
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d.ls1, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d.ls2, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d.mu, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(const_cast<double*>(h->d.rateA), r.data(), r.size() * sizeof(double), hipMemcpyHostToDevice));
    return EXAADMM_OK;
}

extern "C" int exaadmm_download_membuf(exaadmm_handle* h, double* host, int64_t nrow, int64_t nline)
{
    REQUIRE(h && host, "null argument");
    REQUIRE(nrow == 31 && nline == h->L, "membuf must be 31 x nline");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> a(h->Lp), b(h->Lp), c(h->Lp), r(h->Lp);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(a.data(), h->d.ls1, a.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), h->d.ls2, b.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c.data(), h->d.mu, c.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r.data(), h->d.rateA, r.size() * sizeof(double), hipMemcpyDeviceToHost));
    memset(host, 0, sizeof(double) * 31 * (size_t)h->L);
    for (int i = 0; i < h->L; i++) {
        const int q = h->perm[i];
        host[24 + 31 * (size_t)i] = a[q];
        host[25 + 31 * (size_t)i] = b[q];
        host[26 + 31 * (size_t)i] = c[q];
        host[28 + 31 * (size_t)i] = r[q];
    }
    return EXAADMM_OK;
}

extern "C" int exaadmm_set_load(exaadmm_handle* h, const double* Pd, const double* Qd, int64_t nbus)
{
    REQUIRE(h && Pd && Qd, "null argument");
    REQUIRE(nbus == h->B, "nbus mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(const_cast<double*>(h->d.Pd), Pd, (size_t)h->B * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(const_cast<double*>(h->d.Qd), Qd, (size_t)h->B * sizeof(double), hipMemcpyHostToDevice));
    h->busc_dirty = true;   // (the per-unit load sits among the bus constants)
    return EXAADMM_OK;
}

extern "C" int exaadmm_set_pg_bounds(exaadmm_handle* h, const double* lo, const double* hi, int64_t ngen)
{
    REQUIRE(h && lo && hi, "null argument");
    REQUIRE(ngen == h->G, "ngen mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(const_cast<double*>(h->d.pgmin), lo, (size_t)h->G * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(const_cast<double*>(h->d.pgmax), hi, (size_t)h->G * sizeof(double), hipMemcpyHostToDevice));
    return EXAADMM_OK;
}

// init_solution! is model construction (runs once): computed on the host in the reference
// layout, then uploaded.  acopf_init_solution_cpu.jl:1-45 / acopf_model.jl:61-64,87-89.
static int attach_init_data(exaadmm_handle* h, const exaadmm_grid* g)
{
    InitData* id = &h->init;
    id->Vmin.assign(g->Vmin, g->Vmin + h->B);
    id->Vmax.assign(g->Vmax, g->Vmax + h->B);
    id->pgmin.assign(g->pgmin, g->pgmin + h->G);
    id->pgmax.assign(g->pgmax, g->pgmax + h->G);
    id->qgmin.assign(g->qgmin, g->qgmin + h->G);
    id->qgmax.assign(g->qgmax, g->qgmax + h->G);
    const double* Ysrc[8] = {g->YffR, g->YffI, g->YftR, g->YftI, g->YttR, g->YttI, g->YtfR, g->YtfI};
    for (int k = 0; k < 8; k++) id->Y[k].assign(Ysrc[k], Ysrc[k] + h->L);
    id->brBus.resize(2 * (size_t)h->L);
    for (size_t i = 0; i < id->brBus.size(); i++) {
        id->brBus[i] = (int)(g->brBusIdx[i] - g->index_base);
        REQUIRE(id->brBus[i] >= 0 && id->brBus[i] < h->B, "brBusIdx out of range");
    }
    return EXAADMM_OK;
}

extern "C" int exaadmm_init_solution(exaadmm_handle* h, double rho_pq, double rho_va)
{
    REQUIRE(h, "null handle");
    InitData* id = &h->init;
    HIPCHK(hipSetDevice(h->device));
    const int64_t nvar = h->nvar;
    std::vector<double> zero(nvar, 0.0), v(nvar, 0.0), rho(nvar, rho_pq);
    for (int g = 0; g < h->G; g++) {
        v[2 * g] = 0.5 * (id->pgmin[g] + id->pgmax[g]);
        v[2 * g + 1] = 0.5 * (id->qgmin[g] + id->qgmax[g]);
    }
    for (int l = 0; l < h->L; l++) {
        const int fr = id->brBus[2 * l], to = id->brBus[2 * l + 1];
        const double wij0 = (id->Vmax[fr] * id->Vmax[fr] + id->Vmin[fr] * id->Vmin[fr]) / 2;
        const double wji0 = (id->Vmax[to] * id->Vmax[to] + id->Vmin[to] * id->Vmin[to]) / 2;
        const double wR0 = sqrt(wij0 * wji0);
        const size_t p = 2 * (size_t)h->G + 8 * (size_t)l;
        v[p] = id->Y[0][l] * wij0 + id->Y[2][l] * wR0;
        v[p + 1] = -id->Y[1][l] * wij0 - id->Y[3][l] * wR0;
        v[p + 2] = id->Y[4][l] * wji0 + id->Y[6][l] * wR0;
        v[p + 3] = -id->Y[5][l] * wji0 - id->Y[7][l] * wR0;
        v[p + 4] = wij0;
        v[p + 5] = wji0;
        for (int k = 4; k < 8; k++) rho[p + k] = rho_va;
    }
    int rc;
    const int zeros[] = {EXAADMM_U_CURR, EXAADMM_L_CURR, EXAADMM_RD, EXAADMM_RP, EXAADMM_Z_CURR,
                         EXAADMM_Z_PREV, EXAADMM_LZ, EXAADMM_AX_PLUS_BY};
    for (int w : zeros)
        if ((rc = exaadmm_upload_solution(h, w, zero.data(), nvar))) return rc;
    if ((rc = exaadmm_upload_solution(h, EXAADMM_V_CURR, v.data(), nvar))) return rc;
    if ((rc = exaadmm_upload_solution(h, EXAADMM_RHO, rho.data(), nvar))) return rc;
    // membuf (rows 25-27: line-limit multipliers and penalty; row 29: rateA) and pgmin_curr / pgmax_curr are
    // NOT touched: the reference fills them in the ModelAcopf constructor (acopf_model.jl:61-64,87-89) --
    // here exaadmm_create -- and init_solution! leaves them alone, which solve_mpacopf's warm start and the
    // rolling horizon rely on.  exaadmm_upload_membuf / exaadmm_set_pg_bounds reset them explicitly.
    if (h->T > 1) {
        // init_solution! of the multi-period model (mpacopf_init_solution_cpu.jl:1-20): ramp blocks zero,
        // rho = rho_pq, phat_{t-1} = pbar_{t-1} (= v of the previous period), s = p_t - phat = -phat
        // (u is zero here).  gen_membuf (multiplier, penalty) is NOT touched, as in the reference.
        const size_t n = (size_t)h->T * h->g1;
        std::vector<double> z0(n, 0.0), ru(n, 0.0), rs(n, 0.0), rr(n, rho_pq);
        for (int t = 1; t < h->T; t++)
            for (int g = 0; g < h->g1; g++) {
                const size_t i = (size_t)t * h->g1 + g;
                ru[i] = v[2 * ((size_t)(t - 1) * h->g1 + g)];
                rs[i] = 0.0 - ru[i];
            }
        double* zs[] = {h->d.r_z, h->d.r_zp, h->d.r_l, h->d.r_lz, h->d.r_rp, h->d.r_rd, h->d.r_ax};
        for (double* q : zs) HIPCHK(hipMemcpy(q, z0.data(), n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d.r_u, ru.data(), n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d.r_s, rs.data(), n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d.r_rho, rr.data(), n * sizeof(double), hipMemcpyHostToDevice));
        h->busc_dirty = true;
    }
    h->residual_vectors_valid = true;
    return EXAADMM_OK;
}

// ------------------------------------------------------------------ multi-period model (ModelMpacopf)
// One TIME-EXPANDED handle: the T periods are block-diagonal copies of the network (generator
// t*ngen+i, branch t*nline+j, bus t*nbus+k) with their own loads, so every kernel of the
// single-period path covers all periods in one launch; the ramp coupling adds the 3-variable
// generator subproblem (k_gen_ramp), the ramp terms of the bus update and the ramp block of the
// elementwise steps.  Reference: models/mpacopf/mpacopf_model.jl:40-112.
extern "C" int exaadmm_mp_create(const exaadmm_grid* g, const exaadmm_params* p, int T, const double* Pd, const double* Qd,
                                 double ramp_ratio, int device_id, exaadmm_handle** out)
{
    REQUIRE(g && out && Pd && Qd, "null argument");
    REQUIRE(T >= 1, "need at least one period");
    REQUIRE(g->nline_owned == 0 || g->nline_owned == g->nline, "a multi-period model cannot be a partition");
    const int64_t G = g->ngen, L = g->nline, B = g->nbus;
    REQUIRE((2 * G + 8 * L) * (int64_t)T < (int64_t)2000000000, "time-expanded grid too large for 32-bit device indices");
    const int base = (int)g->index_base;
    auto rep = [&](const double* src, int64_t n) {
        std::vector<double> o((size_t)(n * T));
        for (int t = 0; t < T; t++) std::copy(src, src + n, o.begin() + (size_t)t * n);
        return o;
    };
    std::vector<double> pgmin = rep(g->pgmin, G), pgmax = rep(g->pgmax, G), qgmin = rep(g->qgmin, G), qgmax = rep(g->qgmax, G);
    std::vector<double> c2 = rep(g->c2, G), c1 = rep(g->c1, G), c0 = rep(g->c0, G);
    std::vector<double> Y[8];
    const double* Ysrc[8] = {g->YffR, g->YffI, g->YftR, g->YftI, g->YttR, g->YttI, g->YtfR, g->YtfI};
    for (int k = 0; k < 8; k++) Y[k] = rep(Ysrc[k], L);
    std::vector<double> frvm = rep(g->FrVmBound, 2 * L), tovm = rep(g->ToVmBound, 2 * L), frva = rep(g->FrVaBound, 2 * L),
                        tova = rep(g->ToVaBound, 2 * L), rate = rep(g->rateA, L);
    std::vector<double> vmin = rep(g->Vmin, B), vmax = rep(g->Vmax, B), yshr = rep(g->YshR, B), yshi = rep(g->YshI, B);
    std::vector<double> pd(Pd, Pd + (size_t)B * T), qd(Qd, Qd + (size_t)B * T);
    std::vector<int64_t> brbus((size_t)(2 * L * T));
    for (int t = 0; t < T; t++)
        for (int64_t k = 0; k < 2 * L; k++) brbus[(size_t)(t * 2 * L + k)] = g->brBusIdx[k] + (int64_t)t * B;
    auto csr = [&](const int64_t* st, const int64_t* ix, int64_t nnz, std::vector<int64_t>& S, std::vector<int64_t>& I) {
        S.assign((size_t)(B * T + 1), 0);
        I.assign((size_t)(nnz * T), 0);
        for (int t = 0; t < T; t++) {
            for (int64_t k = 0; k < B; k++) S[(size_t)(t * B + k)] = st[k] + (int64_t)t * nnz;
            for (int64_t k = 0; k < nnz; k++) I[(size_t)(t * nnz + k)] = ix[k] + (int64_t)t * nnz;
        }
        S[(size_t)(B * T)] = base + (int64_t)T * nnz;
    };
    REQUIRE(g->FrStart[B] - base == L && g->ToStart[B] - base == L && g->GenStart[B] - base == G, "CSR arrays must cover every component");
    std::vector<int64_t> frs, fri, tos, toi, gs, gi;
    csr(g->FrStart, g->FrIdx, L, frs, fri);
    csr(g->ToStart, g->ToIdx, L, tos, toi);
    csr(g->GenStart, g->GenIdx, G, gs, gi);

    exaadmm_grid e;
    memset(&e, 0, sizeof(e));
    e.ngen = G * T; e.nline = L * T; e.nbus = B * T;
    e.index_base = g->index_base;
    e.baseMVA = g->baseMVA;
    e.pgmin = pgmin.data(); e.pgmax = pgmax.data(); e.qgmin = qgmin.data(); e.qgmax = qgmax.data();
    e.c2 = c2.data(); e.c1 = c1.data(); e.c0 = c0.data();
    e.YffR = Y[0].data(); e.YffI = Y[1].data(); e.YftR = Y[2].data(); e.YftI = Y[3].data();
    e.YttR = Y[4].data(); e.YttI = Y[5].data(); e.YtfR = Y[6].data(); e.YtfI = Y[7].data();
    e.FrVmBound = frvm.data(); e.ToVmBound = tovm.data(); e.FrVaBound = frva.data(); e.ToVaBound = tova.data();
    e.rateA = rate.data();
    e.brBusIdx = brbus.data();
    e.Pd = pd.data(); e.Qd = qd.data(); e.Vmin = vmin.data(); e.Vmax = vmax.data(); e.YshR = yshr.data(); e.YshI = yshi.data();
    e.FrStart = frs.data(); e.FrIdx = fri.data(); e.ToStart = tos.data(); e.ToIdx = toi.data();
    e.GenStart = gs.data(); e.GenIdx = gi.data();

    exaadmm_handle* h = nullptr;
    int rc = exaadmm_create(&e, p, device_id, &h);
    if (rc) return rc;
    // the bus-sorted branch order keeps every period contiguous (bus ids are period-major)
    for (int t = 0; t < T; t++)
        for (int64_t b = 0; b < L; b++) {
            const int q = h->perm[(size_t)(t * L + b)];
            if (q < t * L || q >= (t + 1) * L) {
                set_err("internal: branch permutation crosses a period boundary");
                exaadmm_destroy(h);
                return EXAADMM_ERR_INVALID;
            }
        }
    h->T = T; h->g1 = (int)G; h->l1 = (int)L; h->b1 = (int)B;
    h->nvar_ctrl = (T == 1) ? (2 * G + 8 * L) : (2 * G + 8 * L + G);   // mpacopf_model.jl:99-104
    Dev& d = h->d;
    d.T = T; d.g1 = (int)G; d.l1 = (int)L; d.b1 = (int)B;
#define TRYM(x) do { if ((rc = (x))) { exaadmm_destroy(h); return rc; } } while (0)
    const size_t n = (size_t)std::max<int64_t>(G * T, 1);
    double** rv[] = {&d.r_u, &d.r_s, &d.r_z, &d.r_zp, &d.r_l, &d.r_lz, &d.r_rho, &d.r_rp, &d.r_rd, &d.r_ax, &d.r_mu, &d.r_xi};
    for (auto pp : rv) {
        TRYM(dev_alloc(h, pp, n));
        if (hipMemset(*pp, 0, n * sizeof(double)) != hipSuccess) { set_err("hipMemset failed"); exaadmm_destroy(h); return EXAADMM_ERR_HIP; }
    }
    {
        std::vector<double> rr(n, 0.0), cc(n, 0.0);
        for (size_t i = 0; i < (size_t)(G * T); i++) {
            rr[i] = ramp_ratio * pgmax[i];   // acopf_model.jl:66-67
            cc[i] = c0[i];
        }
        double *drr, *dcc;
        TRYM(dev_alloc(h, &drr, n));
        TRYM(dev_alloc(h, &dcc, n));
        TRYM(upload_vec(h, drr, rr));
        TRYM(upload_vec(h, dcc, cc));
        d.ramp_rate = drr;
        d.c0 = dcc;
    }
    // (chunk, period, plane) launch shape of the elementwise steps; partial sums per block
    h->mp_nbx = std::min(256, std::max(1, (int)((std::max(G, L) + EW_BLOCK - 1) / EW_BLOCK)));
    TRYM(dev_alloc(h, &d.partials, (size_t)T * MP_PLANES * h->mp_nbx * 4));
#undef TRYM
    h->dev_dirty = true;
    *out = h;
    return EXAADMM_OK;
}

extern "C" int exaadmm_mp_periods(exaadmm_handle* h) { return h ? h->T : -1; }
extern "C" int64_t exaadmm_mp_nvar(exaadmm_handle* h) { return h ? h->nvar_ctrl : -1; }

extern "C" int exaadmm_mp_update_x_gen(exaadmm_handle* h, int major_iter)
{
    REQUIRE(h && major_iter >= 1, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    launch_gen(h, nullptr, major_iter);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

static double* ramp_ptr(exaadmm_handle* h, int which)
{
    switch (which) {
        case EXAADMM_U_CURR: return h->d.r_u;
        case EXAADMM_L_CURR: return h->d.r_l;
        case EXAADMM_RHO: return h->d.r_rho;
        case EXAADMM_RD: return h->d.r_rd;
        case EXAADMM_RP: return h->d.r_rp;
        case EXAADMM_Z_CURR: return h->d.r_z;
        case EXAADMM_Z_PREV: return h->d.r_zp;
        case EXAADMM_LZ: return h->d.r_lz;
        case EXAADMM_AX_PLUS_BY: return h->d.r_ax;
        case EXAADMM_RAMP_S: return h->d.r_s;
        case EXAADMM_RAMP_MU: return h->d.r_mu;
        case EXAADMM_RAMP_XI: return h->d.r_xi;
        default: return nullptr;
    }
}

extern "C" int exaadmm_mp_download_ramp(exaadmm_handle* h, int which, double* host, int64_t n)
{
    REQUIRE(h && host, "null argument");
    REQUIRE(h->T > 1, "not a multi-period handle");
    REQUIRE(n == (int64_t)h->T * h->g1, "ramp vectors have T x ngen entries (period-major, block 0 unused)");
    double* src = ramp_ptr(h, which);
    REQUIRE(src, "unknown ramp vector id");
    HIPCHK(hipSetDevice(h->device));
    if (which == EXAADMM_RP || which == EXAADMM_RD || which == EXAADMM_AX_PLUS_BY) {
        int rc = materialize_if_needed(h);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return EXAADMM_OK;
}

extern "C" int exaadmm_mp_upload_ramp(exaadmm_handle* h, int which, const double* host, int64_t n)
{
    REQUIRE(h && host, "null argument");
    REQUIRE(h->T > 1, "not a multi-period handle");
    REQUIRE(n == (int64_t)h->T * h->g1, "ramp vectors have T x ngen entries (period-major, block 0 unused)");
    double* dst = ramp_ptr(h, which);
    REQUIRE(dst, "unknown ramp vector id");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(dst, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    if (which == EXAADMM_RHO) h->busc_dirty = true;
    return EXAADMM_OK;
}

// ------------------------------------------------------------------ launches
static inline dim3 grid_for(int n, int block) { return dim3((unsigned)std::max(1, (n + block - 1) / block)); }

static void launch_gen(exaadmm_handle* h, const Ctrl* c, int major_iter)
{
    if (h->G <= 0) return;
    if (h->T > 1)
        hipLaunchKernelGGL(k_gen_ramp, grid_for(h->G, 64), dim3(64), 0, h->stream, h->d, h->bo, h->baseMVA, major_iter, c);
    else
        hipLaunchKernelGGL(k_generator, grid_for(h->G, 64), dim3(64), 0, h->stream, h->d, h->baseMVA, c);
}
template <int MODE>
static void launch_mp_elem(exaadmm_handle* h, double beta, const Ctrl* c)
{
    hipLaunchKernelGGL((k_mp_elem<MODE>), dim3(h->mp_nbx, h->T, MP_PLANES), dim3(EW_BLOCK), 0, h->stream, h->d, beta,
                       h->par.MAX_MULTIPLIER, c);
}
// 64 branches per wavefront (full waves) is the measured optimum at every size: spreading a small grid over more, emptier
// waves shortens no wave's serial chain by much and costs more than it saves (profiles/archive_r01_r02/r02_lanes_per_wave.txt).
static int branch_blocks(const exaadmm_handle* h) { return (int)grid_for(h->L, BR_BLOCK).x; }
static int gen_blocks(const exaadmm_handle* h) { return (h->G > 0) ? (int)grid_for(h->G, BR_BLOCK).x : 0; }

// the Dev copy kernels read through a pointer (rare: after create / attach)
static void flush_dev(exaadmm_handle* h)
{
    if (h->dev_dirty) {
        (void)hipMemcpyAsync(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice, h->stream);
        h->dev_dirty = false;
    }
}
// per-bus constants of the bus update (rho sums, pivots; per-unit load and shunt): recomputed after rho or the load changed
// (never inside a stream capture: callers flush first)
static void flush_bus_const(exaadmm_handle* h)
{
    if (h->busc_dirty) {
        hipLaunchKernelGGL(k_bus_const, grid_for(h->B, 256), dim3(256), 0, h->stream, h->d, h->d_busc, h->baseMVA);
        h->busc_dirty = false;
    }
}

static void launch_branch(exaadmm_handle* h, int major_iter, const Ctrl* c, bool with_generators = false)
{
    flush_dev(h);
    const int nb = branch_blocks(h);
    const int ng = with_generators ? gen_blocks(h) : 0;
    if (h->T > 1)
        hipLaunchKernelGGL((k_branch_t<true>), dim3(nb + ng), dim3(BR_BLOCK), 0, h->stream, h->d_dev, h->bo, h->par.scale, major_iter, const_cast<Ctrl*>(c), ng, -1, h->baseMVA);
    else
        hipLaunchKernelGGL((k_branch_t<false>), dim3(nb + ng), dim3(BR_BLOCK), 0, h->stream, h->d_dev, h->bo, h->par.scale, major_iter, const_cast<Ctrl*>(c), ng, -1, h->baseMVA);
}
static void launch_bus(exaadmm_handle* h, const Ctrl* c)
{
    if (h->bus_lane_kernel)  // A/B switch (EXAADMM_BUS_KERNEL=lane): one lane per bus, everything derived in place
        hipLaunchKernelGGL(k_bus, grid_for(h->B, 64), dim3(64), 0, h->stream, h->d, h->baseMVA, c);
    else {
        flush_bus_const(h);
        hipLaunchKernelGGL((k_bus_group_t<false>), grid_for(h->B, BUS_BLOCK / BUS_W), dim3(BUS_BLOCK), 0, h->stream, c, (Ctrl*)nullptr, (const double*)h->d.rec,
                           h->d.EndCount, h->d.SlotCode, h->d.busc, h->d.B, h->d, h->baseMVA);
    }
}
static void launch_fused_zl(exaadmm_handle* h, double beta, const Ctrl* c)
{
    if (h->T > 1)
        launch_mp_elem<MP_FUSED_ZL>(h, beta, c);
    else
        hipLaunchKernelGGL(k_fused_zl, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d, beta, c, (int*)nullptr);
}
// single-period: d_sums receives the four SUMS OF SQUARES; multi-period: the four NORMS (max over periods)
static void launch_final(exaadmm_handle* h, Ctrl* c, double* out = nullptr)
{
    if (h->T > 1)
        hipLaunchKernelGGL(k_final_reduce_mp, dim3(1), dim3(EW_BLOCK), 0, h->stream, h->d.partials, MP_PLANES * h->mp_nbx,
                           h->T, out ? out : h->d_sums, c);
    else
        hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(EW_BLOCK), 0, h->stream, h->d.partials, h->ew_blocks,
                           out ? out : h->d_sums, c, 0, 1);
}

static int sync_check(exaadmm_handle* h, bool force)
{
    HIPCHK(hipGetLastError());
    if (force || !h->async_mode) HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

static int fetch_sums(exaadmm_handle* h, double out[4], bool take_sqrt)
{
    HIPCHK(hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    int rc = sync_check(h, true);
    if (rc) return rc;
    if (h->T > 1) take_sqrt = false;  // the multi-period reduction delivers norms
    for (int j = 0; j < 4; j++) out[j] = take_sqrt ? sqrt(h->h_sums[j]) : h->h_sums[j];
    return 0;
}

// a partition with peers (dist_native.h): one grouped send / receive per peer + the unpack kernel
static bool dist_has_peers(const exaadmm_handle* h);
static int dist_check_exchange(exaadmm_handle* h);
static bool dist_graph_allowed(const exaadmm_handle* h);
static void dist_graph_refused(exaadmm_handle* h);
static void dist_count_batch(exaadmm_handle* h, bool graph, int exchanges);
static void dist_sample_exchange(exaadmm_handle* h, bool arm);
static int dist_exchange(exaadmm_handle* h, const Dev& dv, bool with_v, Ctrl* c, double* out, int sums_mode, hipStream_t stream);

// ------------------------------------------------------------------ step-wise plugin surface
extern "C" int exaadmm_inner_prestep(exaadmm_handle* h)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->T > 1)
        launch_mp_elem<MP_COPY_Z>(h, 0.0, nullptr);
    else
        hipLaunchKernelGGL(k_copy_z, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_update_x_gen(exaadmm_handle* h)
{
    REQUIRE(h, "null handle");
    REQUIRE(h->T == 1, "multi-period handle: use exaadmm_mp_update_x_gen(h, major_iter)");
    HIPCHK(hipSetDevice(h->device));
    launch_gen(h, h->ctrl_mode ? h->d_ctrl : nullptr);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_update_x_line(exaadmm_handle* h, int major_iter)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    launch_branch(h, major_iter, h->ctrl_mode ? h->d_ctrl : nullptr);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_update_x(exaadmm_handle* h, int major_iter)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    // generators (multi-period: ramp subproblems) ride in the branch launch: one kernel boundary less
    launch_branch(h, major_iter, h->ctrl_mode ? h->d_ctrl : nullptr, true);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_update_xbar(exaadmm_handle* h)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    launch_bus(h, h->ctrl_mode ? h->d_ctrl : nullptr);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_update_z(exaadmm_handle* h, double beta)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->T > 1)
        launch_mp_elem<MP_UPDATE_Z>(h, beta, nullptr);
    else
        hipLaunchKernelGGL(k_update_z, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d, beta);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_update_l(exaadmm_handle* h, double beta)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->T > 1)
        launch_mp_elem<MP_UPDATE_L>(h, beta, nullptr);
    else
        hipLaunchKernelGGL(k_update_l, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d, beta);
    return sync_check(h);
}

extern "C" int exaadmm_update_lz(exaadmm_handle* h, double beta)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->T > 1)
        launch_mp_elem<MP_UPDATE_LZ>(h, beta, nullptr);
    else
        hipLaunchKernelGGL(k_update_lz, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d, beta, h->par.MAX_MULTIPLIER);
    return sync_check(h);
}

extern "C" int exaadmm_update_residual(exaadmm_handle* h, double out[4])
{
    REQUIRE(h && out, "null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->T > 1)
        launch_mp_elem<MP_RESIDUAL>(h, 0.0, nullptr);
    else
        hipLaunchKernelGGL(k_residual, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d);
    launch_final(h, nullptr);
    h->residual_vectors_valid = true;
    return fetch_sums(h, out, true);
}

extern "C" int exaadmm_update_zl_partial(exaadmm_handle* h, double beta, double* dev_out4)
{
    REQUIRE(h && dev_out4, "null argument");
    REQUIRE(h->T == 1, "partition helpers are not available on a multi-period handle");
    HIPCHK(hipSetDevice(h->device));
    launch_fused_zl(h, beta, h->ctrl_mode ? h->d_ctrl : nullptr);
    launch_final(h, nullptr, dev_out4);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

extern "C" int exaadmm_sumsq_z_partial(exaadmm_handle* h, double* dev_out4)
{
    REQUIRE(h && dev_out4, "null argument");
    REQUIRE(h->T == 1, "partition helpers are not available on a multi-period handle");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_sumsq_z, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d);
    launch_final(h, nullptr, dev_out4);
    return sync_check(h);
}

extern "C" int exaadmm_norm_z(exaadmm_handle* h, double* out)
{
    REQUIRE(h && out, "null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->T > 1)
        launch_mp_elem<MP_SUMSQ_Z>(h, 0.0, nullptr);
    else
        hipLaunchKernelGGL(k_sumsq_z, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d);
    launch_final(h, nullptr);
    if (dist_has_peers(h)) {   // a partition: the owned elements' sum of squares, added over the ranks in rank order
        hipLaunchKernelGGL(k_halo_pack_sums, dim3(1), dim3(64), 0, h->stream, h->d, h->d_sums, (const Ctrl*)nullptr);
        int rc = dist_exchange(h, h->d, false, nullptr, h->d_sums, 0, h->stream);
        if (rc) return rc;
    }
    double s[4];
    int rc = fetch_sums(h, s, true);
    if (rc) return rc;
    if ((rc = dist_check_exchange(h))) return rc;
    *out = s[2];
    return EXAADMM_OK;
}

extern "C" int exaadmm_objective(exaadmm_handle* h, double* out)
{
    // admm_poststep (acopf_admm_prepoststep_cpu.jl:42-45): sum_g c2 (M pg)^2 + c1 (M pg) + c0
    REQUIRE(h && out, "null argument");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> pg(std::max(h->G, 1));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->G > 0) HIPCHK(hipMemcpy(pg.data(), h->d.u, (size_t)h->G * sizeof(double), hipMemcpyDeviceToHost));
    double obj = 0.0;
    for (int g = 0; g < h->G; g++) {
        const double p = h->baseMVA * pg[g];
        obj += h->c2[g] * (p * p) + h->c1[g] * p + h->c0[g];
    }
    *out = obj;
    return EXAADMM_OK;
}

extern "C" int exaadmm_branch_stats(exaadmm_handle* h, double* avg_auglag_it, double* avg_minor_it)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    std::vector<int> a(h->L), m(h->L);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(a.data(), h->d.st_auglag, (size_t)h->L * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(m.data(), h->d.st_minor, (size_t)h->L * sizeof(int), hipMemcpyDeviceToHost));
    double sa = 0.0, sm = 0.0;
    for (int i = 0; i < h->L; i++) {
        sa += a[i];
        sm += a[i] > 0 ? (double)m[i] / a[i] : 0.0;
    }
    if (avg_auglag_it) *avg_auglag_it = sa / h->L;
    if (avg_minor_it) *avg_minor_it = sm / h->L;
    return EXAADMM_OK;
}

extern "C" int exaadmm_download_branch_counters(exaadmm_handle* h, int32_t* a, int32_t* m, int32_t* t, int64_t nline)
{
    REQUIRE(h && nline == h->L, "nline mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> tmp(h->L);
    int32_t* outs[3] = {a, m, t};
    const int* srcs[3] = {h->d.st_auglag, h->d.st_minor, h->d.st_ticks};
    for (int q = 0; q < 3; q++) {
        if (!outs[q]) continue;
        HIPCHK(hipMemcpy(tmp.data(), srcs[q], (size_t)h->L * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < h->L; b++) outs[q][b] = tmp[h->perm[b]];   // back to reference branch order
    }
    return EXAADMM_OK;
}

// per tile of 64 branches IN STORAGE ORDER (= per wavefront of the branch launch / the resident loop): the largest number of solver-loop
// trips any of its lanes made in the last x-update -- what the wavefront executed.  bench.py / scripts/wave_trips.py: instructions per
// wavefront trip = SQ_INSTS_VALU of the launch / the sum of these
extern "C" int exaadmm_download_tile_trips(exaadmm_handle* h, int32_t* out, int64_t ntiles)
{
    REQUIRE(h && out && ntiles == (int64_t)(h->Lp / BR_BLOCK), "buffer must hold one entry per tile of 64 branches");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> t(h->Lp);
    HIPCHK(hipMemcpy(t.data(), h->d.st_ticks, (size_t)h->Lp * sizeof(int), hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < ntiles; q++) {
        int m = 0;
        for (int k = 0; k < BR_BLOCK; k++) {
            const int b = (int)q * BR_BLOCK + k;
            if (b < h->L) m = std::max(m, t[b]);
        }
        out[q] = m;
    }
    return EXAADMM_OK;
}

// Resident wavefronts per SIMD the branch kernel's register / LDS footprint allows (occupancy API, 64-thread
// blocks: blocks per CU / 4 SIMDs).  Diagnostics for bench.py's roofline block.
extern "C" int exaadmm_branch_kernel_occupancy(exaadmm_handle* h, int* waves_per_simd)
{
    REQUIRE(h && waves_per_simd, "null argument");
    HIPCHK(hipSetDevice(h->device));
    int blocks = 0;
    if (h->T > 1)
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_branch_t<true>, BR_BLOCK, 0));
    else
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (k_branch_t<false, true>), BR_BLOCK, 0));
    *waves_per_simd = blocks / 4;
    return EXAADMM_OK;
}

#if defined(EXA_PROFILE_PHASES)
// diagnostic build only (-DEXA_PROFILE_PHASES, libexaadmm_hip_prof.so): per-branch cycle counters of the solver phases, per-wavefront timelines
extern "C" int exaadmm_debug_phase_cycles(exaadmm_handle* h, unsigned long long* out, int64_t n)
{
    REQUIRE(h && out && n == 13 * (int64_t)h->Lp, "buffer must hold 13 x Lp counters");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->d.st_cyc, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return EXAADMM_OK;
}

extern "C" int exaadmm_debug_wave_timeline(exaadmm_handle* h, unsigned long long* out, int64_t n)
{
    REQUIRE(h && out && n == 2 * (int64_t)h->Lp, "buffer must hold 128 entries per wavefront (2 x Lp)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->d.st_tl, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d.st_tl, 0, (size_t)n * sizeof(unsigned long long)));
    return EXAADMM_OK;
}
#endif

// ------------------------------------------------------------------ the device-resident inner loop
// Two launches per iteration (kernels.h):
//     branch launch k+1 : [ z, l, residual terms of iteration k, per lane ] -> [ x-update k+1 ]      (k_branch_t<false, true>)
//     bus launch    k+1 : [ bus update k+1 ]  ||  one extra block: residual sums of iteration k       (k_bus_group_t<true>, `red`)
// and, for a partition with peers, ONE exchange: the v entries the bus update has just produced for the local end of
// every cut branch + this rank's four residual sums -> grouped ncclSend / ncclRecv -> k_halo_unpack (scatter, add the
// sums in rank order, advance the loop control).  The loop test of iteration k is known one launch late; the iterate is
// double-buffered by iteration parity so that the stopping iteration's state survives its speculative successor.
// (option "mp_spec" = 0: A/B switch, the multi-period model keeps the four-launch loop)
static bool spec_eligible(const exaadmm_handle* h) { return h->spec_on && !h->ctrl_mode; }

// resident loop: every wavefront of the launch must hold a SIMD for the whole launch (one wavefront per SIMD)
static int resident_capacity(exaadmm_handle* h)
{
    // what the occupancy calculator says THIS device holds of THIS kernel (registers: one wavefront per SIMD; LDS: 20 KB per
    // block), minus a margin of 8; computed once per handle.  A process that runs under a CU mask does not own the CUs the
    // device properties report: no resident loop there.
    if (h->resident_cap < 0) {
        h->resident_cap = 0;
        hipDeviceProp_t prop;
        int blocks = 0;
        const bool masked = getenv("HSA_CU_MASK") != nullptr || getenv("ROC_GLOBAL_CU_MASK") != nullptr;
        h->resident_cap2 = 0;
        if (!masked && hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_resident_t<1>, BR_BLOCK, 0) == hipSuccess)
                h->resident_cap = std::max(0, std::min(blocks, 4) * prop.multiProcessorCount - 8);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_resident_t<2>, BR_BLOCK, 0) == hipSuccess)
                h->resident_cap2 = std::max(0, std::min(blocks, 4) * prop.multiProcessorCount - 8);
            // (the forms a partition with peers runs must be resident as well: the smaller figure counts)
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (k_resident_t<1, false, true>), BR_BLOCK, 0) == hipSuccess)
                h->resident_cap = std::min(h->resident_cap, std::max(0, std::min(blocks, 4) * prop.multiProcessorCount - 8));
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (k_resident_t<1, true, true>), BR_BLOCK, 0) == hipSuccess)
                h->resident_cap = std::min(h->resident_cap, std::max(0, std::min(blocks, 4) * prop.multiProcessorCount - 8));
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (k_resident_t<2, false, true>), BR_BLOCK, 0) == hipSuccess)
                h->resident_cap2 = std::min(h->resident_cap2, std::max(0, std::min(blocks, 4) * prop.multiProcessorCount - 8));
        }
        if (h->opt.resident_cap_limit > 0) {
            h->resident_cap = std::min(h->resident_cap, h->opt.resident_cap_limit);
            h->resident_cap2 = std::min(h->resident_cap2, h->opt.resident_cap_limit);
        }
        (void)hipGetLastError();
        // a cooperative launch makes the runtime check co-residency against its own occupancy figure and refuse what does
        // not fit (A/B switch, option "resident_coop" = 0: plain launch; measured in profiles/archive_r03_r04/r04_resident_hardening.txt)
        int coop = 0;
        h->resident_coop = h->opt.resident_coop != 0 &&
                           hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, h->device) == hipSuccess && coop != 0;
        (void)hipGetLastError();
    }
    return h->resident_cap;
}

static int resident_mode(exaadmm_handle* h);

// 8-byte-word copies of several buffers in one launch (the resident loop's guard snapshot: one kernel, no host loop of memcpys)
struct CopyList {
    const unsigned long long* src[12];
    unsigned long long* dst[12];
    unsigned int words[12];
    int n;
};
__global__ void k_copy_list(CopyList c)
{
    for (int q = 0; q < c.n; q++)
        for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.words[q]; i += gridDim.x * blockDim.x) c.dst[q][i] = c.src[q][i];
}
// What a call that runs the resident loop does before its first launch, in ONE launch (it used to be a host-to-device copy of the
// control block, two memsets and the guard's copy kernel, each with its own launch gap: ~50 us in front of every run, 5 % of a
// 20-iteration run at 13 659 buses): the guard snapshot, `zero[0 .. nzero)` = 0 (arrive | ready | contrib, decided, stop | fail),
// and -- if `ctrl` is not null -- the control block the loop starts from.
__global__ void k_loop_begin(CopyList c, int* zero, unsigned int nzero, Ctrl* ctrl, Ctrl init)
{
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int q = 0; q < c.n; q++)
        for (unsigned int i = tid; i < c.words[q]; i += nt) c.dst[q][i] = c.src[q][i];
    for (unsigned int i = tid; i < nzero; i += nt) zero[i] = 0;
    if (ctrl != nullptr && tid == 0) *ctrl = init;
}

// everything an inner loop changes that lives in the PRIMARY buffers (where every loop starts and ends): u, v, z, z_prev, l,
// the line-limit multipliers and penalty, the per-branch counters
static int resident_guard(exaadmm_handle* h, bool restore, const Ctrl* begin = nullptr)
{
    int rc;
    if (h->guard.empty()) {
        for (int q = 0; q < 5; q++) {
            double* p = nullptr;
            if ((rc = dev_alloc(h, &p, (size_t)h->N))) return rc;
            h->guard.push_back(p);
        }
        for (int q = 0; q < 3; q++) {
            double* p = nullptr;
            if ((rc = dev_alloc(h, &p, (size_t)h->Lp))) return rc;
            h->guard.push_back(p);
        }
        for (int q = 0; q < 3; q++) {
            int* p = nullptr;
            if ((rc = dev_alloc(h, &p, (size_t)h->Lp))) return rc;   // (Lp is a multiple of 64: whole 8-byte words)
            h->guard.push_back(p);
        }
    }
    void* prim[11] = {h->d.u, h->d.v, h->d.z, h->d.zp, h->d.l, h->d.ls1, h->d.ls2, h->d.mu, h->d.st_auglag, h->d.st_minor, h->d.st_ticks};
    CopyList c;
    c.n = 11;
    for (int q = 0; q < 11; q++) {
        c.src[q] = (const unsigned long long*)(restore ? h->guard[q] : prim[q]);
        c.dst[q] = (unsigned long long*)(restore ? prim[q] : h->guard[q]);
        c.words[q] = (unsigned int)(q < 5 ? h->N : (q < 8 ? h->Lp : h->Lp / 2));
    }
    if (!restore && h->in_selfcheck) c.n = 0;   // (a leg of the self check: the guard keeps the iterate of the call that is being checked)
    if (begin != nullptr || !restore) {
        // the snapshot rides in the call's first launch, with the reset of the hand-over words and (run_fixed) the control block
        const ResidentSync& r = h->rsync;
        Ctrl init;
        memset(&init, 0, sizeof(init));
        if (begin) init = *begin;
        hipLaunchKernelGGL(k_loop_begin, dim3(std::min(1024, std::max(1, h->N / 256))), dim3(256), 0, h->stream, c, r.arrive,
                           (unsigned int)(2 * (size_t)h->B + 4), begin ? h->d_ctrl : (Ctrl*)nullptr, init);
    } else
        hipLaunchKernelGGL(k_copy_list, dim3(std::min(1024, std::max(1, h->N / 256))), dim3(256), 0, h->stream, c);
    HIPCHK(hipGetLastError());
    return 0;
}

static int resident_prepare(exaadmm_handle* h, const Ctrl* begin = nullptr)
{
    ResidentSync& r = h->rsync;
    int rc;
    if (!r.arrive) {
        // arrive[B] | ready[B] | contrib, decided, stop | fail: ONE int buffer; everything but `fail` is reset in front of every
        // launch, `fail` once per call (a later launch's reset must not erase what an earlier one reported)
        int* ints = nullptr;
        if ((rc = dev_alloc(h, &ints, 2 * (size_t)h->B + 4 + 8))) return rc;   // (+ 8: diagnostics behind `fail`)
        r.arrive = ints;
        r.ready = ints + h->B;
        r.contrib = ints + 2 * (size_t)h->B;
        r.decided = r.contrib + 1;
        r.stop = r.contrib + 2;
        r.fail = r.contrib + 3;
        const size_t nwork = (size_t)branch_blocks(h) + gen_blocks(h);
        for (int q = 0; q < 2; q++) {
            if ((rc = dev_alloc(h, &r.terms[q], 4 * h->nrec_slots))) return rc;
            if ((rc = dev_alloc(h, &r.busres[q], 4 * (size_t)h->B))) return rc;
            if ((rc = dev_alloc(h, &r.partials[q], 4 * nwork))) return rc;
            HIPCHK(hipMemsetAsync(r.terms[q], 0, 4 * h->nrec_slots * sizeof(double), h->stream));
        }
    }
    if (resident_mode(h) == 1 && !h->rheavy.list) {
        // wavefronts of their own for the branches with the most solver trips: as many as the launch leaves SIMDs idle (at most 512)
        const int spare = resident_capacity(h) - (branch_blocks(h) + gen_blocks(h) + 1);
        h->heavy_max = std::max(0, std::min(512, spare));
        int* t = nullptr;
        if ((rc = dev_alloc(h, &t, (size_t)std::max(1, h->heavy_max) + (size_t)h->Lp))) return rc;
        h->rheavy.list = t;
        h->rheavy.flag = t + std::max(1, h->heavy_max);
        double* hu = nullptr;
        if ((rc = dev_alloc(h, &hu, (3 * 16 + 32) * (size_t)std::max(1, h->heavy_max)))) return rc;
        h->rheavy.u = hu;
        h->rheavy.init = hu + 2 * 16 * (size_t)std::max(1, h->heavy_max);
        h->rheavy.adopt = hu + 3 * 16 * (size_t)std::max(1, h->heavy_max);
        int* rq = nullptr;
        if ((rc = dev_alloc(h, &rq, (size_t)std::max(1, h->heavy_max) + 2))) return rc;
        HIPCHK(hipMemsetAsync(rq, 0, ((size_t)std::max(1, h->heavy_max) + 2) * sizeof(int), h->stream));
        h->rheavy.req = rq;
        h->rheavy.cnt = rq + std::max(1, h->heavy_max);
    }
    if (!r.last3) {
        if ((rc = dev_alloc(h, &r.last3, (size_t)h->Lp))) return rc;
        HIPCHK(hipMemsetAsync(r.last3, 0, (size_t)h->Lp * sizeof(int), h->stream));
    }
    if (resident_mode(h) == 2 && !h->rtiles.tiles) {
        const int nb = branch_blocks(h), ng = gen_blocks(h);
        int* t = nullptr;
        h->rtiles.nb = nb;
        h->rtiles.ng = ng;
        h->rtiles.wg = (ng + 1) / 2;
        h->rtiles.wb = h->resident_cap2 - h->rtiles.wg - 1;
        if ((rc = dev_alloc(h, &t, 2 * (size_t)h->rtiles.wb + (size_t)nb))) return rc;   // the table, then the tiles' keys
        h->rtiles.tiles = t;
    }
    h->h_base[1] = 0;
    h->resident_zeroed = true;   // (the call's first launch finds its hand-over words reset: enqueue_resident)
    // the iterate the call starts from -- what a failed batch falls back to --, the reset of arrive .. fail, the control block: one launch
    return resident_guard(h, false, begin);
}

// one launch = `n` iterations of the loop (the batch); the loop control is read and advanced inside (k_resident)
// iterations one resident launch may cover (see exaadmm_run_fixed): the lanes pack the launch-relative iteration index into 32-bit words
// next to a branch number (index << 20) and into mailbox sequence numbers ((epoch << 12) | index, 0xfff reserved for the agreement kernel)
constexpr int RESIDENT_LAUNCH_MAX = 1024;
static int enqueue_resident(exaadmm_handle* h, int n, int loop_iter0)
{
    REQUIRE(n >= 1 && n <= RESIDENT_LAUNCH_MAX, "internal: a resident launch covers 1 .. 1024 iterations");
    const int nb = branch_blocks(h), ng = gen_blocks(h);
    if (!h->resident_zeroed)   // (everything but `fail`: a later launch's reset must not erase what an earlier one reported)
        HIPCHK(hipMemsetAsync(h->rsync.arrive, 0, (2 * (size_t)h->B + 3) * sizeof(int), h->stream));
    h->resident_zeroed = false;
    h->resident_launches += 1;
#if defined(EXA_TEST_HOOKS)
    // test build only (exaadmm_test_hook "fail_resident_launch"): the n-th resident launch of a handle reports a failed wait
    if (h->hook_fail_launch != 0 && h->hook_fail_launch == h->resident_launches) HIPCHK(hipMemsetAsync(h->rsync.fail, 0x01, 1, h->stream));
#endif
    const int mode = resident_mode(h);
    int grid = nb + ng + 1;
    if (mode == 2) {
        // which tiles share a wavefront in this launch: from the solver-trip counters of the last x-updates (both buffer sets)
        const ResidentTiles& t = h->rtiles;
        int* const keys = const_cast<int*>(t.tiles) + 2 * (size_t)t.wb;   // (behind the table)
        hipLaunchKernelGGL(k_tile_keys, dim3((t.nb + 3) / 4), dim3(256), 0, h->stream, (const int*)h->sets[0].st[2], (const int*)h->sets[1].st[2], h->L, t.nb, keys,
                           dist_has_peers(h) ? h->rpeers.cut_code : (const int*)nullptr);
        hipLaunchKernelGGL(k_tile_assign, dim3(1), dim3(1024), 2 * (size_t)t.nb * sizeof(int), h->stream, (const int*)keys, t.nb, t.wb, const_cast<int*>(t.tiles));
        grid = t.wb + t.wg + 1;
    }
    ResidentHeavy rh = h->rheavy;
    rh.n = 0;
    // Heavy wavefronts (kernels.h) serve launches that CONTINUE an inner loop beyond its 64th iteration: the branches they take
    // are chosen from the trip counters of the iterations before the launch, and at the start of an inner loop (the penalties
    // restart, the slowest branch changes from iteration to iteration) those say nothing -- a launch from there runs the kernel
    // without them (profiles/archive_r03_r04/r04_heavy_wavefronts.txt: +8.6 % in the steady state, -2.7 % over a whole solve if always on).
    const int heavy_from = (h->opt.heavy_split > 0) ? std::min(HEAVY_FROM, h->opt.heavy_split) : HEAVY_FROM;
    const bool heavy_rule = (h->force_heavy >= 0) ? (h->force_heavy != 0) : (loop_iter0 >= heavy_from && h->opt.heavy != 0);
    const bool peers = dist_has_peers(h);
    if (mode == 1 && heavy_rule && std::min(h->heavy_max, h->opt.heavy_max) > 0) {
        // which branches get a wavefront of their own in this launch: from the solver-trip counters of the last x-updates (both buffer sets)
        rh.n = std::min(h->heavy_max, h->opt.heavy_max);
        hipLaunchKernelGGL(k_heavy_select, dim3(1), dim3(h->opt.heavy_select_small != 0 ? 64 : 1024), 0, h->stream, (const int*)h->sets[0].st[2], (const int*)h->sets[1].st[2], h->L, h->Lp, rh.n,
                           const_cast<int*>(rh.list), const_cast<int*>(rh.flag), (const double*)h->d.z, (const double*)h->d.l, 2 * h->Gp, const_cast<double*>(rh.init),
                           peers ? h->rpeers.cut_code : (const int*)nullptr, (const int*)h->rsync.last3, h->resident_iters, HEAVY_WINDOW, rh.req, rh.cnt);
        rh.adopt_thr = h->opt.heavy_adopt;
        grid += rh.n;
    }
    h->form_tiles = mode;
    h->form_heavy = rh.n;
    h->form_grid = grid;
    h->form_launches += 1;
    h->form_used |= (rh.n > 0) ? 2u : 1u;
    ResidentPeers rp = h->rpeers;
    if (peers) {
        h->p2p_epoch = (h->p2p_epoch + 1) & 0x7ffff;   // (the same count on every rank: they make the same calls)
        rp.epoch = h->p2p_epoch;
#if defined(EXA_TEST_HOOKS)
        rp.delay_ticks = h->hook_p2p_delay_ticks;
#endif
    }
    // (a partition: the ranks wait for each other -- patiently -- in front of the launch, not inside it: kernels.h k_p2p_rendezvous)
    if (peers) hipLaunchKernelGGL(k_p2p_rendezvous, dim3(1), dim3(64), 0, h->stream, rp, h->rsync.fail);
    const void* const kernel = peers ? (mode == 2 ? (const void*)k_resident_t<2, false, true> : (rh.n > 0 ? (const void*)k_resident_t<1, true, true> : (const void*)k_resident_t<1, false, true>))
                                     : (mode == 2 ? (const void*)k_resident_t<2> : (rh.n > 0 ? (const void*)k_resident_t<1, true> : (const void*)k_resident_t<1>));
    const Dev *d0 = h->d_dspec[0][0], *d1 = h->d_dspec[1][0];
    double scale = h->par.scale, base = h->baseMVA;
    h->rsync.stamp0 = h->resident_iters;
    h->resident_iters = (h->resident_iters + n) & 0x03ffffff;   // (a wrap once in 67 million iterations costs one launch its history)
    void* args[] = {(void*)&d0, (void*)&d1, (void*)&h->bo, (void*)&scale, (void*)&h->d_ctrl, (void*)&ng, (void*)&base, (void*)&n, (void*)&h->rsync, (void*)&h->rtiles, (void*)&rh, (void*)&rp};
    bool launched = false;
    if (h->resident_coop) {
        const hipError_t e = hipLaunchCooperativeKernel(kernel, dim3(grid), dim3(BR_BLOCK), args, 0, h->stream);
        launched = (e == hipSuccess);
        if (!launched) {
            (void)hipGetLastError();
            if (e != hipErrorCooperativeLaunchTooLarge) h->resident_coop = false;   // (not offered after all: plain launches from here on)
            else if (!peers) return 1;   // the runtime says the grid is not co-resident: the caller takes the two-launch loop
            else {
                // a partition must not leave its peers waiting: the refusal becomes a failed launch that the agreement kernel reports to everybody
                HIPCHK(hipMemsetAsync(h->rsync.fail, 0x01, 1, h->stream));
                launched = true;
            }
        }
    }
    if (!launched) {
        HIPCHK(hipLaunchKernel(kernel, dim3(grid), dim3(BR_BLOCK), args, 0, h->stream));
    }
    if (peers) hipLaunchKernelGGL(k_p2p_agree, dim3(1), dim3(64), 0, h->stream, rp, h->rsync.fail);
    HIPCHK(hipGetLastError());
    return 0;
}

// assume_mailboxes: the answer a partition would get once its peers' mailboxes are mapped (what exaadmm_dist_p2p_export tells the peers)
static int resident_mode_if(exaadmm_handle* h, bool assume_mailboxes)
{
#if defined(EXA_PROFILE_PHASES)
    // (the diagnostic build's per-wavefront timelines belong to k_branch_t; option "prof_resident" = 1: phase clocks of the resident loop's lanes)
    if (h->opt.prof_resident == 0) return 0;
#endif
    if (h->opt.resident == 0 || h->force_two_launch) return 0;
    const bool peers = dist_has_peers(h);
    if (!((peers ? (h->resident_peers_ok && (h->p2p_ready || assume_mailboxes) && h->opt.dist_p2p != 0) : h->resident_ok) && h->d.rec != nullptr && h->T == 1 && h->resident_downgrades == 0 &&
          spec_eligible(h) && h->stream == h->own_stream))
        return 0;
    const int nb = branch_blocks(h), ng = gen_blocks(h);
    if (nb + ng + 1 <= resident_capacity(h)) return 1;
    if (h->opt.resident_tiles == 0) return 0;
    const int wg = (ng + 1) / 2, wb = h->resident_cap2 - wg - 1;
    return (wb > 0 && nb > wb && nb <= 2 * wb) ? 2 : 0;
}
static int resident_mode(exaadmm_handle* h) { return resident_mode_if(h, false); }
static bool resident_eligible(exaadmm_handle* h) { return resident_mode(h) != 0; }

// The failure flag of the resident launches travels with the download that ends a batch (no round trip of its own):
// resident_check_enqueue before that download's synchronisation, resident_failed after it.
static int resident_check_enqueue(exaadmm_handle* h)
{
    HIPCHK(hipMemcpyAsync(&h->h_base[1], h->rsync.fail, sizeof(int), hipMemcpyDeviceToHost, h->stream));   // (pinned)
    return 0;
}
static bool resident_failed(const exaadmm_handle* h) { return h->h_base[1] != 0; }

// A resident batch ran into a wait cap (a wavefront of the launch was not resident: the GPU is shared with other work), or
// the runtime refused the cooperative launch: bring the iterate the call started from back and keep the two-launch loop on
// this handle from here on.  The caller then repeats its loop; the call returns what it would have returned.
static int resident_downgrade(exaadmm_handle* h)
{
    int rc;
    if ((rc = sync_check(h, true))) return rc;
    if ((rc = resident_guard(h, true))) return rc;
    h->resident_downgrades += 1;
    if (h->h_base && h->h_base[1] != 0)   // (a downgrade after a self check keeps that check's message)
        set_err("resident launch did not complete on device %d (wait code %d: 1 a bus / decision of this GPU -- shared GPU? --, 2 a cut branch's twin on a peer, 3 a peer's "
            "residual sums, 4 reported by a peer at the end of the launch, 5 a twin in the launch's last step, 8 a peer that did not reach the launch within the minute the ranks give each other): the call was repeated with two launches per iteration",
                h->device, h->h_base[1]);
    if (h->par.verbose > 0) fprintf(stderr, "exaadmm: %s\n", g_err.c_str());
    return sync_check(h, true);
}

static int interior_blocks(const exaadmm_handle* h) { return h->n_interior / BR_BLOCK; }   // whole blocks in front of the first cut branch

static int resident_mode(exaadmm_handle* h);
static int spec_prepare(exaadmm_handle* h)
{
    int rc;
    // (a partition with peers on the two-launch loop runs two launches ahead of its stop test: three sets; the resident loop of a
    //  partition whose peers' mailboxes are mapped keeps the one-iteration speculation of the single-GPU loops: two)
    const int want = (dist_has_peers(h) && resident_mode(h) == 0) ? 3 : 2;
    // alternate buffer sets: one for the two-set loop, a second one (and alternates of z, l; a sink for the z_prev the kernels
    // write) once the handle has peers
    for (int q = 0; q < want - 1; q++) {
        exaadmm_handle::IterSet& a = h->alt[q];
        if (a.u) continue;
        if ((rc = dev_alloc(h, &a.u, (size_t)h->N))) return rc;
        if ((rc = dev_alloc(h, &a.v, (size_t)h->N))) return rc;
        if ((rc = dev_alloc(h, &a.ls1, (size_t)h->Lp))) return rc;
        if ((rc = dev_alloc(h, &a.ls2, (size_t)h->Lp))) return rc;
        if ((rc = dev_alloc(h, &a.mu, (size_t)h->Lp))) return rc;
        for (auto& c : a.st) {
            if ((rc = dev_alloc(h, &c, (size_t)h->Lp))) return rc;
            HIPCHK(hipMemsetAsync(c, 0, (size_t)h->Lp * sizeof(int), h->stream));
        }
        // padding entries (never written by the branch kernel) must agree in all sets
        HIPCHK(hipMemcpyAsync(a.u, h->d.u, (size_t)h->N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(a.v, h->d.v, (size_t)h->N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemsetAsync(a.ls1, 0, (size_t)h->Lp * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(a.ls2, 0, (size_t)h->Lp * sizeof(double), h->stream));
        HIPCHK(hipMemsetAsync(a.mu, 0, (size_t)h->Lp * sizeof(double), h->stream));
    }
    const size_t nramp = (size_t)std::max(h->T * h->g1, 1);
    if (h->T > 1 && !h->alt[0].r_u) {
        exaadmm_handle::IterSet& a = h->alt[0];
        double** rv[4] = {&a.r_u, &a.r_s, &a.r_mu, &a.r_xi};
        const double* src[4] = {h->d.r_u, h->d.r_s, h->d.r_mu, h->d.r_xi};
        for (int q = 0; q < 4; q++) {
            if ((rc = dev_alloc(h, rv[q], nramp))) return rc;
            HIPCHK(hipMemcpyAsync(*rv[q], src[q], nramp * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    if (want == 3 && !h->zp_scratch) {
        for (int q = 0; q < 2; q++) {
            if ((rc = dev_alloc(h, &h->alt[q].z, (size_t)h->N))) return rc;
            if ((rc = dev_alloc(h, &h->alt[q].l, (size_t)h->N))) return rc;
            HIPCHK(hipMemcpyAsync(h->alt[q].z, h->d.z, (size_t)h->N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->alt[q].l, h->d.l, (size_t)h->N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        if ((rc = dev_alloc(h, &h->zp_scratch, (size_t)h->N))) return rc;
    }
    // more branch blocks than the chip has SIMDs, no peers: the blocks take the tiles heaviest first (Dev::blk_order, refreshed per batch)
    {
        hipDeviceProp_t prop;
        const bool want_order = h->opt.block_order != 0 && !dist_has_peers(h) && hipGetDeviceProperties(&prop, h->device) == hipSuccess &&
                                branch_blocks(h) + gen_blocks(h) > 4 * prop.multiProcessorCount;
        if (want_order && !h->blk_order) {
            const int nbb = branch_blocks(h);
            if ((rc = dev_alloc(h, &h->blk_order, (size_t)nbb))) return rc;
            if ((rc = dev_alloc(h, &h->blk_keys, (size_t)nbb))) return rc;
            std::vector<int> ident(nbb);
            for (int q = 0; q < nbb; q++) ident[q] = q;
            HIPCHK(hipMemcpyAsync(h->blk_order, ident.data(), (size_t)nbb * sizeof(int), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        h->blk_order_off = !want_order;
    }
    // multi-period model: a block's partial sums have one slot per period its 64 lanes can belong to
    const int part_k = (h->T > 1) ? std::min(h->T, (BR_BLOCK - 1) / std::max(1, std::min(h->l1, std::max(h->g1, 1))) + 2) : 1;
    if (!h->partials_br) {
        if ((rc = dev_alloc(h, &h->partials_br, (size_t)(branch_blocks(h) + gen_blocks(h)) * 4 * part_k))) return rc;
        HIPCHK(hipMemsetAsync(h->partials_br, 0, (size_t)(branch_blocks(h) + gen_blocks(h)) * 4 * part_k * sizeof(double), h->stream));
        for (auto& row : h->d_dspec)
            for (auto& q : row)
                if ((rc = dev_alloc(h, &q, 1))) return rc;
        if (hipHostMalloc((void**)&h->h_base, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) {   // [0] batch base, [1] resident failure flag
            set_err("hipHostMalloc failed");
            return EXAADMM_ERR_HIP;
        }
    }
    flush_bus_const(h);
    const int n = want;
    exaadmm_handle::IterSet prim;
    prim.u = h->d.u; prim.v = h->d.v; prim.ls1 = h->d.ls1; prim.ls2 = h->d.ls2; prim.mu = h->d.mu; prim.z = h->d.z; prim.l = h->d.l;
    prim.st[0] = h->d.st_auglag; prim.st[1] = h->d.st_minor; prim.st[2] = h->d.st_ticks;
    prim.r_u = h->d.r_u; prim.r_s = h->d.r_s; prim.r_mu = h->d.r_mu; prim.r_xi = h->d.r_xi;
    exaadmm_handle::IterSet sets[3];
    for (int q = 0; q < n - 1; q++) sets[q] = h->alt[q];
    sets[n - 1] = prim;
    const int nb = branch_blocks(h), ng = gen_blocks(h), nbi = interior_blocks(h);
    Dev var[3][3], fin[3];
    for (int r = 0; r < n; r++) {
        // launch i (i mod n = r) reads the iterate of launch i - 1 and writes its own set.  The per-branch counters follow the
        // iterate: a launch that the stop test discards (it runs ahead) must not overwrite those of the last accepted x-update;
        // v follows the same rotation: the bus launch of iteration i writes the v the next branch launch reads.
        const exaadmm_handle::IterSet &w = sets[r], &rd = sets[(r + n - 1) % n];
        Dev a = h->d;
        a.u_in = rd.u;      a.u = w.u;
        a.v_in = rd.v;      a.v = w.v;
        a.ls1_in = rd.ls1;  a.ls1 = w.ls1;
        a.ls2_in = rd.ls2;  a.ls2 = w.ls2;
        a.mu_in = rd.mu;    a.mu = w.mu;
        a.st_auglag = w.st[0];  a.st_minor = w.st[1];  a.st_ticks = w.st[2];
        if (n == 3) {
            // the z / l pass in launch i's prologue completes iteration i - 1: it reads z, l of iteration i - 2 and writes those of
            // i - 1 into the set of iteration i - 1 (where the solve then finds them); the z_prev it writes is not kept
            a.z = rd.z;  a.l = rd.l;
            a.z_in = sets[(r + 1) % 3].z;  a.l_in = sets[(r + 1) % 3].l;
            a.zp = h->zp_scratch;
        }
        if (h->T > 1) {   // multi-period model: the ramp subproblems' outputs rotate with the iterate (the bus launch gathers the generators)
            a.r_u_in = rd.r_u;    a.r_u = w.r_u;
            a.r_s_in = rd.r_s;    a.r_s = w.r_s;
            a.r_mu_in = rd.r_mu;  a.r_mu = w.r_mu;
            a.r_xi_in = rd.r_xi;  a.r_xi = w.r_xi;
        }
        a.part_k = part_k;
        a.blk_order = h->blk_order_off ? nullptr : h->blk_order;   // (the whole-grid launch; the split launches of a partition keep block i = tile blk0 + i)
        a.partials_br = h->partials_br;
        a.npartials_br = nb + ng;
        a.blk0 = 0;
        a.gen_part0 = nb;
        var[r][0] = var[r][1] = var[r][2] = a;
        var[r][1].blk_order = var[r][2].blk_order = nullptr;
        var[r][2].blk0 = nbi;
        // the pass that completes a loop's LAST launch (index i, r = (i + 1) mod n: what launch i + 1's prologue would have done):
        // u, v of launch i are its inputs; in place on two sets
        fin[r] = a;
        fin[r].u = const_cast<double*>(a.u_in);
        fin[r].v = const_cast<double*>(a.v_in);
        if (h->T > 1) fin[r].r_u = const_cast<double*>(a.r_u_in);
    }
    bool changed = (h->nsets != n);
    for (int r = 0; r < n && !changed; r++)
        for (int q = 0; q < 3; q++) changed = changed || memcmp(&var[r][q], &h->dspec[r][q], sizeof(Dev)) != 0;
    if (changed) {
        invalidate_graph(h);
        h->nsets = n;
        for (int r = 0; r < n; r++) {
            h->sets[r] = sets[r];
            h->dfin[r] = fin[r];
            for (int q = 0; q < 3; q++) {
                h->dspec[r][q] = var[r][q];
                HIPCHK(hipMemcpyAsync(h->d_dspec[r][q], &h->dspec[r][q], sizeof(Dev), hipMemcpyHostToDevice, h->stream));
            }
        }
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// iteration base + j + 1 of the loop on ONE GPU (j = position inside the batch; base is a multiple of the number of sets)
// Dev::blk_order for the launches that follow: the tiles by the largest solver-trip count of their lanes in the last two iterations, heaviest first
static void enqueue_block_order(exaadmm_handle* h)
{
    if (h->blk_order_off || !h->blk_order || h->nsets < 2) return;
    const int nbb = branch_blocks(h);
    hipLaunchKernelGGL(k_tile_keys, dim3((nbb + 3) / 4), dim3(256), 0, h->stream, (const int*)h->sets[0].st[2], (const int*)h->sets[1].st[2], h->L, nbb, h->blk_keys,
                       (const int*)nullptr);
    hipLaunchKernelGGL(k_block_order, dim3(1), dim3(1024), 0, h->stream, (const int*)h->blk_keys, nbb, h->blk_order);
}

static int enqueue_iteration_spec(exaadmm_handle* h, Ctrl* c, int j, bool sample = false, hipEvent_t* bev = nullptr)
{
    const int r = j % h->nsets;
    const Dev& dv = h->dspec[r][0];
    const int nb = branch_blocks(h), ng = gen_blocks(h);
    if (sample) (void)hipEventRecord(h->tev[0], h->stream);
    if (bev) (void)hipEventRecord(bev[0], h->stream);
    if (h->T > 1)
        hipLaunchKernelGGL((k_branch_t<true, true>), dim3(nb + ng), dim3(BR_BLOCK), 0, h->stream, h->d_dspec[r][0], h->bo, h->par.scale, 0, c,
                           ng, j, h->baseMVA);
    else
        hipLaunchKernelGGL((k_branch_t<false, true>), dim3(nb + ng), dim3(BR_BLOCK), 0, h->stream, h->d_dspec[r][0], h->bo, h->par.scale, 0, c,
                           ng, j, h->baseMVA);
    if (bev) (void)hipEventRecord(bev[1], h->stream);
    if (sample) (void)hipEventRecord(h->tev[1], h->stream);
    if (dv.rec != nullptr && h->T > 1)   // multi-period model: records for the branch ends, gathers for the generators (ramp partners)
        hipLaunchKernelGGL((k_bus_group_t<true, true>), dim3(grid_for(h->B, BUS_BLOCK / BUS_W).x + 1), dim3(BUS_BLOCK), 0, h->stream, (const Ctrl*)c, c, (const double*)dv.rec, dv.EndCount, dv.SlotCode,
                           dv.busc, dv.B, dv, h->baseMVA);
    else if (dv.rec != nullptr)   // the branch launch has pushed the bus records
        hipLaunchKernelGGL((k_bus_group_t<true>), dim3(grid_for(h->B, BUS_BLOCK / BUS_W).x + 1), dim3(BUS_BLOCK), 0, h->stream, (const Ctrl*)c, c, (const double*)dv.rec, dv.EndCount, dv.SlotCode,
                           dv.busc, dv.B, dv, h->baseMVA);
    else
        hipLaunchKernelGGL((k_bus_group_t<false>), dim3(grid_for(h->B, BUS_BLOCK / BUS_W).x + 1), dim3(BUS_BLOCK), 0, h->stream, (const Ctrl*)c, c, (const double*)dv.rec, dv.EndCount, dv.SlotCode,
                           dv.busc, dv.B, dv, h->baseMVA);
    if (sample) {
        (void)hipEventRecord(h->tev[2], h->stream);
        (void)hipEventRecord(h->tev[3], h->stream);
    }
    return 0;
}

// iterations [j0, j0 + n) of a partition WITH PEERS (dist_native.h): the pipelined two-stream form of the same iteration
static int enqueue_dist_iterations(exaadmm_handle* h, Ctrl* c, int j0, int n, bool first_of_loop, hipEvent_t* bev);

// After a loop of `iters` completed iterations the valid iterate sits in set (iters - 1) mod nsets: bring it home so that
// every other entry point keeps using the primary buffers.
static int spec_finish(exaadmm_handle* h, int iters)
{
    if (iters <= 0) return 0;
    const int n = h->nsets, f = (iters - 1) % n;
    const size_t NB = (size_t)h->N * sizeof(double), LB = (size_t)h->Lp * sizeof(double);
    const exaadmm_handle::IterSet &src = h->sets[f], &dst = h->sets[n - 1];
    if (n == 3)   // z_prev of the final iterate = z of the iteration before it (before z itself is overwritten below)
        HIPCHK(hipMemcpyAsync(h->d.zp, h->sets[(f + 2) % 3].z, NB, hipMemcpyDeviceToDevice, h->stream));
    if (f != n - 1) {
        HIPCHK(hipMemcpyAsync(dst.u, src.u, NB, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(dst.v, src.v, NB, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(dst.ls1, src.ls1, LB, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(dst.ls2, src.ls2, LB, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(dst.mu, src.mu, LB, hipMemcpyDeviceToDevice, h->stream));
        for (int q = 0; q < 3; q++)
            HIPCHK(hipMemcpyAsync(dst.st[q], src.st[q], (size_t)h->Lp * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        if (h->T > 1) {
            const size_t RB = (size_t)h->T * h->g1 * sizeof(double);
            HIPCHK(hipMemcpyAsync(dst.r_u, src.r_u, RB, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dst.r_s, src.r_s, RB, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dst.r_mu, src.r_mu, RB, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dst.r_xi, src.r_xi, RB, hipMemcpyDeviceToDevice, h->stream));
        }
        if (n == 3) {
            HIPCHK(hipMemcpyAsync(dst.z, src.z, NB, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dst.l, src.l, NB, hipMemcpyDeviceToDevice, h->stream));
        }
    }
    return 0;
}

// The last enqueued iteration has no successor: nothing would run its z/l pass or reduce its residual sums.  Every
// kernel here returns at once when the loop has already stopped.
static int spec_complete_last(exaadmm_handle* h, int enqueued)
{
    if (enqueued <= 0) return 0;
    const Dev& dv = h->dfin[enqueued % h->nsets];   // u, v, (z, l) of the last enqueued iteration -> its z / l pass
    if (h->T > 1) {   // multi-period model: the per-period pass and reduction of the four-launch loop (both return at once after a stop)
        hipLaunchKernelGGL((k_mp_elem<MP_FUSED_ZL>), dim3(h->mp_nbx, h->T, MP_PLANES), dim3(EW_BLOCK), 0, h->stream, dv, 0.0, h->par.MAX_MULTIPLIER, (const Ctrl*)h->d_ctrl);
        hipLaunchKernelGGL(k_final_reduce_mp, dim3(1), dim3(EW_BLOCK), 0, h->stream, h->d.partials, MP_PLANES * h->mp_nbx, h->T, h->d_sums, h->d_ctrl);
        return 0;
    }
    hipLaunchKernelGGL(k_fused_zl, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, dv, 0.0, h->d_ctrl, &h->d_ctrl->pending);
    if (!dist_has_peers(h)) {
        hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(EW_BLOCK), 0, h->stream, h->d.partials, h->ew_blocks, h->d_sums, h->d_ctrl, 1, 1);
        return 0;
    }
    hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(EW_BLOCK), 0, h->stream, h->d.partials, h->ew_blocks, h->d_sums, h->d_ctrl, 1, 0);
    hipLaunchKernelGGL(k_halo_pack_sums, dim3(1), dim3(64), 0, h->stream, h->d, h->d_sums, (const Ctrl*)h->d_ctrl);
    return dist_exchange(h, h->d, false, h->d_ctrl, h->d_sums, 1, h->stream);
}

constexpr int DIST_BATCH = 12;   // iterations per host visit / per hipGraph of a partition with peers (a multiple of its three buffer sets)

__global__ void k_ctrl_set_base(Ctrl* c, int base) { c->base = base; }

// One full batch of a partition's loop (not the first of a loop: every launch has a z / l pass in front) as a hipGraph -- two
// streams, interior / cut / bus launches, the exchanges, the unpack kernels.  Offered when the bound RCCL proved at connect
// time that it can be captured (dist_probe); a capture that fails anyway leaves the handle on direct enqueue (nothing of it has run).
static bool dist_batch_graph(exaadmm_handle* h)
{
    if (h->spec_graph) return true;
    if (!(h->use_graph && h->stream == h->own_stream && dist_graph_allowed(h))) return false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
    int crc = 0;
    if (e == hipSuccess) {
        crc = enqueue_dist_iterations(h, h->d_ctrl, 0, DIST_BATCH, false, nullptr);
        e = hipStreamEndCapture(h->stream, &g);
    }
    if (e == hipSuccess && crc == 0) e = hipGraphInstantiate(&h->spec_graph, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (e != hipSuccess || crc != 0) {
        (void)hipGetLastError();
        h->spec_graph = nullptr;
        dist_graph_refused(h);
        return false;
    }
    return true;
}

static int resident_selfcheck(exaadmm_handle* h, int next_major_iter, double beta);

static int run_inner_spec(exaadmm_handle* h, int max_inner, Ctrl* hc)
{
    int rc = spec_prepare(h);
    if (rc) return rc;
    const Ctrl ctrl0 = *hc;   // what exaadmm_run_inner has just uploaded (a downgraded resident call starts over from it)
    const bool peers = dist_has_peers(h);
    // iterations per host visit: 8 on one GPU, 12 on a partition with peers
    const int batch = peers ? DIST_BATCH : 8;
    const bool graph_ok = h->use_graph && h->stream == h->own_stream;
    float ms_sample[3] = {0.f, 0.f, 0.f};
    int enq = 0;
    bool first = true;
    const bool resident = resident_eligible(h);
    if (resident && (rc = resident_prepare(h))) return rc;
    bool downgrade = false, completed = false;
    while (enq < max_inner) {
        // resident: ONE launch (and one host visit) per 1 024 iterations -- normally the whole inner loop; the launch ends inside the
        // kernel at the iteration the stop test names, so its length only bounds the host visits (a launch has a fixed cost of
        // 0.15 ms at 13 659 buses, 0.23 ms at 88 207 branches: profiles/archive_r03_r04/r04_resident_launch_fixed_cost.txt)
        // (option heavy_split = S > 0: the first launch of an inner loop ends after S iterations, so that the rest of the loop runs with heavy
        //  wavefronts chosen from those S iterations' history -- one more launch per inner loop, 0.15 ms, taken only when the handle's
        //  previous inner loop ran long enough to promise it back (heavy_split_prev iterations): 15.0 k -> 18.2 k it/s over the whole solve of
        //  the 1 354-bus grid, whose loops run 250 iterations; nothing at 13 659 buses, whose loops run 100.  S = 48: with 24 or 32 the adoption spends
        //  the free heavy wavefronts on the passing stragglers of a loop's first iterations -- 17.1 k)
        // (a partition with peers: every rank must give its launches the same lengths, and the spare SIMDs differ from rank to rank -- there the rule is
        //  "every rank runs the one-tile form", which all ranks read from the same mailbox descriptions, exaadmm_dist_p2p_import; the options and the
        //  length of the previous loop are the same on every rank anyway.  A rank without spare SIMDs then splits its launch for nothing: 0.15 ms.
        //  Round 6 withdrew this for a day: the 3-rank solve of case118 over thread ranks left the single handle's trajectory in 3 of 115 runs with it and
        //  in none of 105 without.  The cause was not the split: a sums exchange between the launches that gave up on a late peer after 0.9 s and handed
        //  the outer loop NaN -- as often without the split once the ranks were a little further apart (profiles/r06_heavy_split_on_partitions.txt))
        const bool split = resident && h->opt.heavy_split > 0 && h->opt.heavy != 0 && resident_mode(h) == 1 && enq < h->opt.heavy_split &&
                           (h->last_inner_len < 0 || h->last_inner_len >= h->opt.heavy_split_prev) &&   // (a handle's first loop: one launch risked)
                           (peers ? h->peers_one_tile : h->heavy_max > 0);
        // (a batch that another one follows must be a multiple of the number of buffer sets long: every launch starts on set 0)
        const int split_len = ((h->opt.heavy_split + h->nsets - 1) / h->nsets) * h->nsets;
        const int nb = std::min(resident ? (split && enq < split_len ? split_len - enq : RESIDENT_LAUNCH_MAX) : batch, max_inner - enq);
        *h->h_base = enq;   // the stream is idle here (synchronised at the end of every batch)
        HIPCHK(hipMemcpyAsync(&h->d_ctrl->base, h->h_base, sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (resident) {
            // one launch per batch; a batch that meets the stop ends inside the kernel, at the reference's iteration.  (The three
            // step timers of the reference cannot be told apart inside one launch: the sampled time of the first batch goes to
            // time_x_update.)
            if (first) (void)hipEventRecord(h->tev[0], h->stream);
            if ((rc = enqueue_resident(h, nb, enq)) < 0) return rc;
            if (rc > 0) {   // the runtime refused the cooperative launch
                downgrade = true;
                break;
            }
            if (first) {
                (void)hipEventRecord(h->tev[1], h->stream);
                (void)hipEventRecord(h->tev[2], h->stream);
                (void)hipEventRecord(h->tev[3], h->stream);
            }
            if ((rc = resident_check_enqueue(h))) return rc;
            if (enq + nb >= max_inner) {
                // no batch can follow: the z / l pass and the reduction of the last iteration (they return at once if the loop stopped
                // earlier) ride in front of this batch's ONE synchronisation
                if ((rc = spec_complete_last(h, enq + nb))) return rc;
                completed = true;
            }
        } else if (peers && !first && nb == batch && dist_batch_graph(h)) {
            HIPCHK(hipGraphLaunch(h->spec_graph, h->stream));
            dist_count_batch(h, true, batch);
        } else if (!peers && !first && graph_ok && nb == batch) {
            enqueue_block_order(h);
            if (!h->spec_graph) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
                for (int j = 0; j < batch; j++) (void)enqueue_iteration_spec(h, h->d_ctrl, j);
                HIPCHK(hipStreamEndCapture(h->stream, &g));
                hipError_t e = hipGraphInstantiate(&h->spec_graph, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e != hipSuccess) {
                    set_err("hipGraphInstantiate failed: %s", hipGetErrorString(e));
                    return EXAADMM_ERR_HIP;
                }
            }
            HIPCHK(hipGraphLaunch(h->spec_graph, h->stream));
        } else if (peers) {
            dist_sample_exchange(h, true);
            if ((rc = enqueue_dist_iterations(h, h->d_ctrl, 0, nb, first, nullptr))) return rc;
            dist_count_batch(h, false, 0);
        } else {
            enqueue_block_order(h);
            for (int j = 0; j < nb; j++)
                if ((rc = enqueue_iteration_spec(h, h->d_ctrl, j, first && j == 0))) return rc;
        }
        enq += nb;
        HIPCHK(hipMemcpyAsync(hc, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
        if ((rc = sync_check(h, true))) return rc;
        if (resident && resident_failed(h)) {   // checked after EVERY batch: the iterate of a failed batch is not a state of the algorithm
            downgrade = true;
            break;
        }
        if (peers) dist_sample_exchange(h, false);
        if (first) {
            for (int q = 0; q < 3; q++)   // (a partition with peers interleaves its steps on two streams: no per-step sample there)
                if (peers || hipEventElapsedTime(&ms_sample[q], h->tev[q], h->tev[q + 1]) != hipSuccess) ms_sample[q] = 0.f;
            if (resident) ms_sample[0] /= (float)std::max(1, std::min(nb, std::max(1, hc->inner)));   // per iteration of the sampled batch
            first = false;
        }
        if (hc->done) break;   // (a partition: every rank sees the same sums, added in the same order -> the same decision)
    }
    if (!downgrade) h->last_inner_len = hc->inner;
    if (!downgrade) {
        if (!completed) {
            if ((rc = spec_complete_last(h, enq))) return rc;
            HIPCHK(hipMemcpyAsync(hc, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
            if ((rc = sync_check(h, true))) return rc;
        }
        if ((rc = spec_finish(h, hc->inner))) return rc;
        if (resident && h->form_used != 0) {
            // first use of a resident kernel form on this handle: the loop's result is only returned once 8 more iterations from it
            // came out bit for bit the same with the resident and with the two-launch loop (resident_selfcheck)
            if ((rc = sync_check(h, true))) return rc;
            const int verdict = resident_selfcheck(h, hc->inner + 1, hc->beta);
            if (verdict < 0) return verdict;
            downgrade = verdict > 0;
        }
    }
    if (downgrade) {
        // the inner loop starts over from the iterate it was entered with, on the two-launch loop (same results, bit for bit:
        // tests/test_gpu_resident.py); the handle keeps that loop
        if ((rc = resident_downgrade(h))) return rc;
        *hc = ctrl0;
        HIPCHK(hipMemcpyAsync(h->d_ctrl, hc, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return run_inner_spec(h, max_inner, hc);
    }
    if ((rc = dist_check_exchange(h))) return rc;   // (the sums exchange that completes a loop's last iteration)
    for (int q = 0; q < 3; q++) h->acc_ms[q] += (double)ms_sample[q] * hc->inner;
    return 0;
}

extern "C" int exaadmm_loop_kind(exaadmm_handle* h, int* kind)
{
    REQUIRE(h && kind, "null argument");
    *kind = !spec_eligible(h) ? 0 : (resident_eligible(h) ? 2 : 1);
    return EXAADMM_OK;
}

// how many times a resident batch of this handle was found not co-resident (the GPU shared with other work), restored and
// repeated with the two-launch loop; after the first the handle stays on that loop (exaadmm_loop_kind says 1)
extern "C" int exaadmm_loop_downgrades(exaadmm_handle* h, int* count)
{
    REQUIRE(h && count, "null argument");
    *count = h->resident_downgrades;
    return EXAADMM_OK;
}

// ------------------------------------------------------------------ fused loops
// the plain loop (multi-period model, EXAADMM_NO_SPEC / EXAADMM_BUS_KERNEL=lane A/B runs): four launches per iteration
static void enqueue_iteration(exaadmm_handle* h, Ctrl* c, bool sample = false)
{
    if (sample) (void)hipEventRecord(h->tev[0], h->stream);
    launch_branch(h, 0, c, true);   // generators (multi-period: ramp subproblems) ride in the same launch
    if (sample) (void)hipEventRecord(h->tev[1], h->stream);
    launch_bus(h, c);
    if (sample) (void)hipEventRecord(h->tev[2], h->stream);
    launch_fused_zl(h, 0.0, c);
    launch_final(h, c);
    if (sample) (void)hipEventRecord(h->tev[3], h->stream);
}

static int64_t nvar_for_tolerances(const exaadmm_handle* h);   // a partition: nvar of the WHOLE grid (dist_native.h)

extern "C" int exaadmm_run_inner(exaadmm_handle* h, int outer, int max_inner, double beta, int* inner_done, double out[4])
{
    REQUIRE(h && outer >= 1 && max_inner >= 1, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    REQUIRE(!dist_has_peers(h) || spec_eligible(h), "a partition with peers runs the speculative loop only (single-period model, EXAADMM_NO_SPEC / EXAADMM_BUS_KERNEL unset)");
    h->form_tiles = h->form_heavy = h->form_launches = h->form_grid = 0;
    h->form_used = 0;
    Ctrl* hc = h->h_ctrl;
    memset(hc, 0, sizeof(Ctrl));
    hc->max_inner = max_inner;
    // (a partition tests every iteration: all ranks must take the same decision in the same iteration)
    hc->check_every = dist_has_peers(h) ? 1 : std::max(1, (int)h->par.check_every);
    hc->eps_pri = sqrt((double)nvar_for_tolerances(h)) / (2500.0 * outer);  // admm_two_level.jl:45
    hc->beta = beta;
    hc->first_iter = 1;
    HIPCHK(hipMemcpyAsync(h->d_ctrl, hc, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // hc is reused as the download target below
    h->residual_vectors_valid = false;
    // Kernels are enqueued in batches; each kernel returns at once when the device-side
    // `done` flag is set, so the break happens at exactly the reference's iteration
    // without a host round trip per iteration.
    const int batch = 8;
    int enq = 0;
    for (auto& e : h->tev)
        if (!e) HIPCHK(hipEventCreate(&e));
    // Full batches after the first are replayed from a hipGraph; the first batch
    // of every call is launched directly with HIP events around its first iteration, which feed
    // the sampled step times (mod.info.time_*_update of the reference).  Capturing needs a real
    // stream: with the caller's null stream (exaadmm_set_stream(NULL)) launches stay direct.
    const bool graph_ok = h->use_graph && h->stream == h->own_stream;
    if (h->dev_dirty) {   // flush the Dev copy outside any capture; it is baked into the graph
        invalidate_graph(h);
        flush_dev(h);
    }
    flush_bus_const(h);
    if (spec_eligible(h)) {
        int rc = run_inner_spec(h, max_inner, hc);
        if (rc) return rc;
        if (inner_done) *inner_done = hc->inner;
        if (out)
            for (int j = 0; j < 4; j++) out[j] = hc->res[j];
        return EXAADMM_OK;
    }
    float ms_sample[3] = {0.f, 0.f, 0.f};
    bool first = true;
    while (enq < max_inner) {
        const int nb = std::min(batch, max_inner - enq);
        if (!first && graph_ok && nb == batch) {
            if (!h->batch_graph) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < batch; i++) enqueue_iteration(h, h->d_ctrl, false);
                HIPCHK(hipStreamEndCapture(h->stream, &g));
                hipError_t e = hipGraphInstantiate(&h->batch_graph, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e != hipSuccess) {
                    set_err("hipGraphInstantiate failed: %s", hipGetErrorString(e));
                    return EXAADMM_ERR_HIP;
                }
            }
            HIPCHK(hipGraphLaunch(h->batch_graph, h->stream));
        } else {
            for (int i = 0; i < nb; i++) enqueue_iteration(h, h->d_ctrl, first && i == 0);
        }
        enq += nb;
        HIPCHK(hipMemcpyAsync(hc, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
        int rc = sync_check(h, true);
        if (rc) return rc;
        if (first) {
            for (int q = 0; q < 3; q++)
                if (hipEventElapsedTime(&ms_sample[q], h->tev[q], h->tev[q + 1]) != hipSuccess) ms_sample[q] = 0.f;
            first = false;
        }
        if (hc->done) break;
    }
    for (int q = 0; q < 3; q++) h->acc_ms[q] += (double)ms_sample[q] * hc->inner;
    if (inner_done) *inner_done = hc->inner;
    if (out)
        for (int j = 0; j < 4; j++) out[j] = hc->res[j];
    return EXAADMM_OK;
}

extern "C" int exaadmm_solve_two_level(exaadmm_handle* h, exaadmm_info* out)
{
    // admm_two_level (src/algorithms/admm_two_level.jl:1-88).  On a partition with peers every rank runs this same
    // function: norms and loop decisions are identical on all ranks (sums added in rank order), the scalar logic is
    // replicated, info.objval is THIS partition's share of the objective (the caller adds the shares).
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    exaadmm_params& par = h->par;
    exaadmm_info& info = h->info;
    const double sqrt_d = sqrt((double)nvar_for_tolerances(h));
    const double OUTER_TOL = sqrt_d * par.outer_eps;
    memset(&info, 0, sizeof(info));
    info.mismatch = INFINITY;
    info.norm_z_prev = info.norm_z_curr = INFINITY;
    par.beta = par.initial_beta;
    info.status = 0;
    h->acc_ms[0] = h->acc_ms[1] = h->acc_ms[2] = 0.0;
    int rc;

    HIPCHK(hipStreamSynchronize(h->stream));
    const auto t0 = std::chrono::steady_clock::now();
    while (info.outer < par.outer_iterlim) {
        info.outer += 1;
        if ((rc = exaadmm_norm_z(h, &info.norm_z_prev))) return rc;  // admm_outer_prestep
        info.inner = 0;
        int done = 0;
        double res[4] = {0, 0, 0, 0};
        if ((rc = exaadmm_run_inner(h, info.outer, par.inner_iterlim, par.beta, &done, res))) return rc;
        info.inner = done;
        info.cumul += done;
        info.primres = res[0];
        info.dualres = res[1];
        info.norm_z_curr = res[2];
        info.mismatch = res[3];
        info.eps_pri = sqrt_d / (2500.0 * info.outer);
        if (par.verbose > 0)
            printf("%8d  %8d  %10.3e  %10.3e  %10.3e  %10.3e  %10.3e  %10.3e  %10.3e\n", info.outer, info.inner,
                   info.primres, info.eps_pri, info.dualres, info.norm_z_curr, info.mismatch, OUTER_TOL, par.beta);
        if (info.mismatch <= OUTER_TOL) {
            info.status = 1;
            break;
        }
        if ((rc = exaadmm_update_lz(h, par.beta))) return rc;
        if (info.norm_z_curr > par.theta * info.norm_z_prev) par.beta = fmin(par.inc_c * par.beta, 1e24);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    info.time_overall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info.time_x_update = 1e-3 * h->acc_ms[0];
    info.time_xbar_update = 1e-3 * h->acc_ms[1];
    info.time_zl_update = 1e-3 * h->acc_ms[2];
    if ((rc = exaadmm_objective(h, &info.objval))) return rc;  // admm_poststep
    exaadmm_branch_stats(h, &info.avg_auglag_it, &info.avg_minor_it);
    if (out) *out = info;
    return EXAADMM_OK;
}

static int run_fixed_impl(exaadmm_handle* h, int iters, int first_major_iter, double beta, double out[4], double* ms_total, double* ms_branch);

extern "C" int exaadmm_run_fixed(exaadmm_handle* h, int iters, int first_major_iter, double beta, double out[4],
                                 double* ms_total, double* ms_branch)
{
    REQUIRE(h && iters >= 1 && first_major_iter >= 1, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    h->form_tiles = h->form_heavy = h->form_launches = h->form_grid = 0;
    h->form_used = 0;
    int rc = run_fixed_impl(h, iters, first_major_iter, beta, out, ms_total, ms_branch);
    if (rc) return rc < 0 ? rc : EXAADMM_ERR_HIP;
    if (h->form_used != 0) {   // first use of a resident kernel form on this handle (resident_selfcheck)
        const int verdict = resident_selfcheck(h, first_major_iter + iters, beta);
        if (verdict < 0) return verdict;
        if (verdict > 0) {
            // the resident loop did not reproduce the two-launch loop on this device: the call's own result is not returned either --
            // back to the iterate it was entered with, and once more with the two-launch loop (which the handle keeps)
            if ((rc = resident_downgrade(h))) return rc;
            rc = run_fixed_impl(h, iters, first_major_iter, beta, out, ms_total, ms_branch);
            if (rc) return rc < 0 ? rc : EXAADMM_ERR_HIP;
        }
    }
    return EXAADMM_OK;
}

// returns 0, a negative status, or (a leg of the self check only) 1: the resident launch reported a failed wait / was refused
static int run_fixed_impl(exaadmm_handle* h, int iters, int first_major_iter, double beta, double out[4], double* ms_total, double* ms_branch)
{
    REQUIRE(!dist_has_peers(h) || spec_eligible(h), "a partition with peers runs the speculative loop only");
    const bool time_branch = (ms_branch != nullptr);
    const size_t need = 2 + (time_branch ? 2 * (size_t)iters : 0);
    while (h->ev.size() < need) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    h->residual_vectors_valid = false;
    flush_dev(h);
    flush_bus_const(h);
    if (spec_eligible(h)) {
        // same kernels as the fused loop, with a control block that never stops
        int rc;
        for (int attempt = 0; attempt < 2; attempt++) {
            // (inside the loop: a partition with peers that falls back from the resident loop changes from two buffer sets to three)
            if ((rc = spec_prepare(h))) return rc;
            Ctrl* hc = h->h_ctrl;
            memset(hc, 0, sizeof(Ctrl));
            hc->max_inner = 0x7fffffff;
            hc->check_every = 1;
            hc->eps_pri = -1.0;
            hc->beta = beta;
            hc->base = first_major_iter - 1;
            hc->first_iter = first_major_iter;
            const bool resident = resident_eligible(h);
            if (resident) {   // (the control block travels in the opening launch)
                if ((rc = resident_prepare(h, hc))) return rc;
            } else
                HIPCHK(hipMemcpyAsync(h->d_ctrl, hc, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));   // (hc is pinned and not written again before the call's final synchronisation)
            HIPCHK(hipEventRecord(h->ev[0], h->stream));
            bool refused = false;
            if (resident) {
                // ONE launch for all `iters` iterations: every wavefront stays resident, a bus is updated by the last lane to arrive at it
                if (time_branch) HIPCHK(hipEventRecord(h->ev[2], h->stream));
                // A launch covers at most RESIDENT_LAUNCH_MAX iterations, the length exaadmm_run_inner's launches have: the words the lanes of a
                // launch pack an iteration index into (adoption requests: index << 20 in 32 bits; mailbox sequence numbers: (epoch << 12) | index)
                // hold 1 024 of them with room to spare, not an unbounded count.  Every rank of a partition cuts the same `iters` the same way.
                const int chunk = std::max(h->nsets, (RESIDENT_LAUNCH_MAX / h->nsets) * h->nsets);
                for (int i0 = 0; i0 < iters && !refused; i0 += chunk) {
                    if (i0 > 0) hipLaunchKernelGGL(k_ctrl_set_base, dim3(1), dim3(1), 0, h->stream, h->d_ctrl, first_major_iter - 1 + i0);
                    if ((rc = enqueue_resident(h, std::min(chunk, iters - i0), first_major_iter - 1 + i0)) < 0) return rc;
                    refused = rc > 0;
                }
                if (time_branch) HIPCHK(hipEventRecord(h->ev[3], h->stream));   // (one launch: its time / iters is reported)
            } else if (dist_has_peers(h)) {
                // batches of 12 like the convergence-tested loop: the first one enqueued directly (its first launch has no z / l
                // pass), full batches after it replayed from the hipGraph when there is one, the rest directly
                for (int i0 = 0; i0 < iters; i0 += DIST_BATCH) {
                    const int n = std::min(DIST_BATCH, iters - i0);
                    if (i0 > 0) hipLaunchKernelGGL(k_ctrl_set_base, dim3(1), dim3(1), 0, h->stream, h->d_ctrl, first_major_iter - 1 + i0);
                    if (i0 > 0 && n == DIST_BATCH && !time_branch && dist_batch_graph(h)) {
                        HIPCHK(hipGraphLaunch(h->spec_graph, h->stream));
                        dist_count_batch(h, true, DIST_BATCH);
                        continue;
                    }
                    if (i0 == 0) dist_sample_exchange(h, true);   // (one exchange of the first, directly enqueued batch is bracketed by events; read after the call's synchronisation)
                    if ((rc = enqueue_dist_iterations(h, h->d_ctrl, 0, n, i0 == 0, time_branch ? &h->ev[2 + 2 * i0] : nullptr))) return rc;
                    dist_count_batch(h, false, 0);
                }
            } else
                for (int i = 0; i < iters; i++) {
                    if ((i & 15) == 0) enqueue_block_order(h);
                    if ((rc = enqueue_iteration_spec(h, h->d_ctrl, i, false, time_branch ? &h->ev[2 + 2 * i] : nullptr))) return rc;
                }
            if (resident && refused) {   // the runtime says the grid is not co-resident
                if (h->in_selfcheck) return 1;
                if ((rc = resident_downgrade(h))) return rc;
                continue;
            }
            if ((rc = spec_complete_last(h, iters))) return rc;
            HIPCHK(hipEventRecord(h->ev[1], h->stream));
            if ((rc = spec_finish(h, iters))) return rc;
            // the failure flag of a resident launch travels with the call's ONE synchronisation (what ran behind a failed launch
            // worked on an iterate that is thrown away)
            if (resident && (rc = resident_check_enqueue(h))) return rc;
            double s[4];
            if ((rc = fetch_sums(h, s, true))) return rc;
            if ((rc = dist_check_exchange(h))) return rc;
            if (resident && resident_failed(h)) {
                // a batch that was not co-resident is not a state of the algorithm: back to the iterate the call was entered
                // with, and once more with the two-launch loop (resident_eligible is false from here on)
                if (h->in_selfcheck) return 1;
                if ((rc = resident_downgrade(h))) return rc;
                continue;
            }
            if (dist_has_peers(h)) dist_sample_exchange(h, false);
            if (out)
                for (int j = 0; j < 4; j++) out[j] = s[j];
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
            if (ms_total) *ms_total = ms;
            if (time_branch && resident) {
                HIPCHK(hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
                *ms_branch = (double)ms / iters;
            } else if (time_branch) {
                double acc = 0.0;
                for (int i = 0; i < iters; i++) {
                    HIPCHK(hipEventElapsedTime(&ms, h->ev[2 + 2 * i], h->ev[3 + 2 * i]));
                    acc += ms;
                }
                *ms_branch = acc / iters;
            }
            return EXAADMM_OK;
        }
        set_err("internal: the two-launch loop was not selected after a resident downgrade");
        return EXAADMM_ERR_HIP;
    }
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    for (int i = 0; i < iters; i++) {
        if (time_branch) HIPCHK(hipEventRecord(h->ev[2 + 2 * i], h->stream));
        launch_branch(h, first_major_iter + i, nullptr, true);
        if (time_branch) HIPCHK(hipEventRecord(h->ev[3 + 2 * i], h->stream));
        launch_bus(h, nullptr);
        launch_fused_zl(h, beta, nullptr);
        launch_final(h, nullptr);
    }
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    double s[4];
    int rc = fetch_sums(h, s, true);
    if (rc) return rc;
    if (out)
        for (int j = 0; j < 4; j++) out[j] = s[j];
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    if (ms_total) *ms_total = ms;
    if (time_branch) {
        double acc = 0.0;
        for (int i = 0; i < iters; i++) {
            HIPCHK(hipEventElapsedTime(&ms, h->ev[2 + 2 * i], h->ev[3 + 2 * i]));
            acc += ms;
        }
        *ms_branch = acc / iters;
    }
    return EXAADMM_OK;
}

// ------------------------------------------------------------------ first-use self check of the resident loops
// The resident kernels hand values from wavefront to wavefront across XCDs; that this works rests on hardware behaviour (sc1
// accesses are performed at the device's coherence point), and round 4 met three silent races there that only bitwise
// comparisons found.  So nothing a resident kernel form has produced on a handle is returned before the form has reproduced the
// two-launch loop ON THIS HANDLE: at the end of the first call that used a form (plain / two-tile: bit 0, heavy wavefronts: bit 1),
// from the iterate E that call produced,
//     E -> SELFCHECK_ITERS iterations, resident loop (the same form)   -> R
//     E -> SELFCHECK_ITERS iterations, two launches per iteration      -> T
// and R, T are compared on the device, word by word: u, v, z, z_prev, l, the line-limit multipliers and penalty, the per-branch
// counters; the four norms on the host, bit for bit.  E is restored.  Returns 0 = equal (the form is marked checked), 1 = a
// difference or a failed leg -- the caller restores the iterate its call STARTED from (the guard snapshot, which the legs leave alone)
// and repeats the call with the two-launch loop, which the handle keeps --, < 0 = error.  Cost: 2 x 8 iterations + four copy launches,
// ~1 ms at 13 659 buses, once per form and handle.  Option "trust_resident" = 1 skips it.
constexpr int SELFCHECK_ITERS = 8;
// What a check establishes is a property of (device, kernel form, launch geometry) -- how this chip's coherence machinery treats the
// hand-over sequence of that kernel at that size -- not of the handle: a form that has passed for a geometry in this process is not
// checked again for another handle of the same geometry (a second solve_acopf of the same case, the periods of a rolling horizon).
static std::mutex g_selfcheck_mutex;
static std::vector<std::array<int, 6>> g_selfcheck_passed;   // device, form, G, L, B, resident capacity
static bool selfcheck_known(const exaadmm_handle* h, int form, bool record)
{
    const std::array<int, 6> key = {h->device, form, h->G, h->L, h->B, h->resident_cap};
    std::lock_guard<std::mutex> lock(g_selfcheck_mutex);
    for (const auto& k : g_selfcheck_passed)
        if (k == key) return true;
    if (record) g_selfcheck_passed.push_back(key);
    return false;
}

__global__ void k_compare_list(CopyList c, unsigned long long* __restrict__ ndiff)
{
    for (int q = 0; q < c.n; q++) {
        unsigned long long cnt = 0;
        for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.words[q]; i += gridDim.x * blockDim.x) cnt += (c.dst[q][i] != c.src[q][i]) ? 1ull : 0ull;
        if (cnt) atomicAdd(&ndiff[q], cnt);
    }
}
#if defined(EXA_TEST_HOOKS)
__global__ void k_flip_ulp(double* p) { *reinterpret_cast<unsigned long long*>(p) += 1ull; }
#endif

static const char* const SELFCHECK_NAMES[12] = {"u", "v", "z", "z_prev", "l", "line-limit multiplier 1", "line-limit multiplier 2", "penalty",
                                               "auglag counters", "minor-iteration counters", "solver-trip counters", "solver-trip counters (other set)"};

// the iterate (+ counters) <-> a snapshot set; cmp != nullptr: compare instead of copying
static int selfcheck_list(exaadmm_handle* h, std::vector<void*>& store, int mode /* 0 save, 1 restore, 2 compare */)
{
    int rc;
    if (store.empty()) {
        for (int q = 0; q < 12; q++) {
            if (q < 8) {
                double* p = nullptr;
                if ((rc = dev_alloc(h, &p, q < 5 ? (size_t)h->N : (size_t)h->Lp))) return rc;
                store.push_back(p);
            } else {
                int* p = nullptr;
                if ((rc = dev_alloc(h, &p, (size_t)h->Lp))) return rc;
                store.push_back(p);
            }
        }
    }
    void* prim[12] = {h->d.u, h->d.v, h->d.z, h->d.zp, h->d.l, h->d.ls1, h->d.ls2, h->d.mu, h->d.st_auglag, h->d.st_minor, h->d.st_ticks, h->sets[0].st[2]};
    CopyList c;
    c.n = 12;
    for (int q = 0; q < 12; q++) {
        c.src[q] = (const unsigned long long*)(mode == 1 ? store[q] : prim[q]);
        c.dst[q] = (unsigned long long*)(mode == 1 ? prim[q] : store[q]);
        c.words[q] = (unsigned int)(q < 5 ? h->N : (q < 8 ? h->Lp : h->Lp / 2));
    }
    const dim3 grid(std::min(1024, std::max(1, h->N / 256)));
    if (mode == 2) {
        HIPCHK(hipMemsetAsync(h->d_chk_diff, 0, 12 * sizeof(unsigned long long), h->stream));
        hipLaunchKernelGGL(k_compare_list, grid, dim3(256), 0, h->stream, c, h->d_chk_diff);
    } else
        hipLaunchKernelGGL(k_copy_list, grid, dim3(256), 0, h->stream, c);
    HIPCHK(hipGetLastError());
    return 0;
}

static int resident_selfcheck(exaadmm_handle* h, int next_major_iter, double beta)
{
    const unsigned used = h->form_used;
    h->form_used = 0;
    if (h->in_selfcheck || h->opt.trust_resident != 0 || h->nsets != 2 || dist_has_peers(h)) return 0;   // (a partition's loop is checked against the single handle by the tests)
    if (!h->checked_resident && selfcheck_known(h, 0, false)) h->checked_resident = true;
    if (!h->checked_heavy && selfcheck_known(h, 1, false)) h->checked_heavy = true;
    const unsigned todo = used & ~((h->checked_resident ? 1u : 0u) | (h->checked_heavy ? 2u : 0u));
    if (todo == 0) return 0;
    int rc;
    if (!h->d_chk_diff && (rc = dev_alloc(h, &h->d_chk_diff, 12))) return rc;
    std::vector<void*>&e_set = h->chk_e, &r_set = h->chk_r;
    const Ctrl keep = *h->h_ctrl;   // (the caller reads its loop's outcome from the pinned block after we return)
    const int form_keep[4] = {h->form_tiles, h->form_heavy, h->form_launches, h->form_grid};
    h->in_selfcheck = true;
    int verdict = 0;
    std::string what;
    if ((rc = selfcheck_list(h, e_set, 0))) goto out_rc;
    for (int form = 0; form < 2 && verdict == 0; form++) {
        if (!(todo & (1u << form))) continue;
        double nr[4] = {0, 0, 0, 0}, nt[4] = {0, 0, 0, 0};
        h->force_heavy = form;
        rc = run_fixed_impl(h, SELFCHECK_ITERS, next_major_iter, beta, nr, nullptr, nullptr);
        h->force_heavy = -1;
        if (rc < 0) goto out_rc;
        h->selfchecks += 1;
        if (rc > 0) {   // the leg itself met a failed wait / a refused launch
            verdict = 1;
            what = "the resident leg of the self check was not co-resident";
        }
#if defined(EXA_TEST_HOOKS)
        if (verdict == 0 && h->hook_flip_ulp != 0) {   // test build only: one ulp on one word of the resident leg's result
            h->hook_flip_ulp = 0;
            hipLaunchKernelGGL(k_flip_ulp, dim3(1), dim3(1), 0, h->stream, h->d.u + 2 * h->Gp + std::min(h->L - 1, 17));
        }
#endif
        if (verdict == 0) {
            if ((rc = selfcheck_list(h, r_set, 0))) goto out_rc;
            if ((rc = selfcheck_list(h, e_set, 1))) goto out_rc;
            h->force_two_launch = true;
            rc = run_fixed_impl(h, SELFCHECK_ITERS, next_major_iter, beta, nt, nullptr, nullptr);
            h->force_two_launch = false;
            if (rc) { rc = rc < 0 ? rc : EXAADMM_ERR_HIP; goto out_rc; }
            if ((rc = selfcheck_list(h, r_set, 2))) goto out_rc;
            unsigned long long nd[12];
            if (hipMemcpyAsync(nd, h->d_chk_diff, sizeof(nd), hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
                set_err("self check: download of the comparison failed");
                rc = EXAADMM_ERR_HIP;
                goto out_rc;
            }
            if ((rc = sync_check(h, true))) goto out_rc;
            char buf[160];
            for (int q = 0; q < 12; q++)
                if (nd[q] != 0) {
                    snprintf(buf, sizeof(buf), "%s%s: %llu word(s)", what.empty() ? "" : ", ", SELFCHECK_NAMES[q], nd[q]);
                    what += buf;
                }
            if (memcmp(nr, nt, sizeof(nr)) != 0) what += what.empty() ? "the residual norms" : ", the residual norms";
            if (!what.empty()) verdict = 1;
        }
        if ((rc = selfcheck_list(h, e_set, 1))) goto out_rc;
        if ((rc = sync_check(h, true))) goto out_rc;
        if (verdict == 0) {
            (form == 0 ? h->checked_resident : h->checked_heavy) = true;
            (void)selfcheck_known(h, form, true);
        }
    }
    rc = 0;
out_rc:
    h->in_selfcheck = false;
    h->force_heavy = -1;
    h->force_two_launch = false;
    *h->h_ctrl = keep;
    h->form_tiles = form_keep[0]; h->form_heavy = form_keep[1]; h->form_launches = form_keep[2]; h->form_grid = form_keep[3];
    if (rc) return rc;
    if (verdict != 0) {
        h->selfcheck_mismatches += 1;
        set_err("resident loop self check: after %d iterations from the same iterate the resident loop and the two-launch loop differ in %s; "
                "the handle keeps the two-launch loop", SELFCHECK_ITERS, what.c_str());
        if (h->par.verbose > 0) fprintf(stderr, "exaadmm: %s\n", g_err.c_str());
    }
    return verdict;
}

extern "C" int exaadmm_resident_selfcheck_stats(exaadmm_handle* h, int* checks, int* mismatches)
{
    REQUIRE(h, "null handle");
    if (checks) *checks = h->selfchecks;
    if (mismatches) *mismatches = h->selfcheck_mismatches;
    return EXAADMM_OK;
}

extern "C" int exaadmm_resident_form(exaadmm_handle* h, int* tiles, int* heavy, int* launches, int* wavefronts)
{
    REQUIRE(h, "null handle");
    if (tiles) *tiles = h->form_launches > 0 ? h->form_tiles : 0;
    if (heavy) *heavy = h->form_launches > 0 ? h->form_heavy : 0;
    if (launches) *launches = h->form_launches;
    if (wavefronts) *wavefronts = h->form_launches > 0 ? h->form_grid : 0;
    return EXAADMM_OK;
}

// ------------------------------------------------------------------ options
namespace {
struct OptKey {
    const char* name;
    int Options::*field;
    int lo, hi;
};
const OptKey OPT_KEYS[] = {
    {"bus_kernel_lane", &Options::bus_kernel_lane, 0, 1}, {"graph", &Options::graph, 0, 1}, {"spec", &Options::spec, 0, 1}, {"rec", &Options::rec, 0, 1},
    {"resident", &Options::resident, 0, 1}, {"resident_tiles", &Options::resident_tiles, 0, 1}, {"heavy", &Options::heavy, 0, 1}, {"heavy_max", &Options::heavy_max, 0, 512},
    {"heavy_adopt", &Options::heavy_adopt, 0, 63}, {"heavy_split", &Options::heavy_split, 0, 1 << 20}, {"heavy_split_prev", &Options::heavy_split_prev, 0, 1 << 20}, {"block_order", &Options::block_order, 0, 1},
    {"dist_overlap", &Options::dist_overlap, 0, 1}, {"dist_p2p", &Options::dist_p2p, 0, 1}, {"trust_resident", &Options::trust_resident, 0, 1},
#if defined(EXA_PROFILE_PHASES)
    {"prof_resident", &Options::prof_resident, 0, 1},   // (diagnostic build only)
#endif
};
#if defined(EXA_TEST_HOOKS)
// switches of the TEST build only (exaadmm_test_hook): what the GPU tests need to reach code paths a product run selects by itself
const OptKey TEST_KEYS[] = {
    {"resident_coop", &Options::resident_coop, 0, 1}, {"resident_cap_limit", &Options::resident_cap_limit, 0, 1 << 20}, {"heavy_select_small", &Options::heavy_select_small, 0, 1},
};
#endif
}  // namespace

// sets one switch of `table`; returns 1 when the key is not in it
template <size_t NK>
static int set_switch(exaadmm_handle* h, const OptKey (&table)[NK], const char* key, int64_t value)
{
    for (const OptKey& k : table) {
        if (strcmp(k.name, key) != 0) continue;
        if (value < k.lo || value > k.hi) {
            set_err("option %s: value %lld out of range [%d, %d]", key, (long long)value, k.lo, k.hi);
            return EXAADMM_ERR_INVALID;
        }
        if (h->opt.*(k.field) == (int)value) return EXAADMM_OK;
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->opt.*(k.field) = (int)value;
        // what the options feed: the bus kernel / loop selection, the records pointer inside every Dev copy, the graphs that bake them in
        h->bus_lane_kernel = h->opt.bus_kernel_lane != 0;
        h->use_graph = h->opt.graph != 0;
        h->spec_on = h->opt.spec != 0 && !h->bus_lane_kernel;   // (the speculative loop needs the lane-group bus kernel: it carries the reduction block)
        h->d.rec = h->opt.rec != 0 ? h->rec_alloc : nullptr;
        h->dev_dirty = true;
        h->resident_cap = -1;   // (resident_coop / resident_cap_limit are looked at when the capacity is computed)
        invalidate_graph(h);
        return EXAADMM_OK;
    }
    return 1;
}

extern "C" int exaadmm_set_option(exaadmm_handle* h, const char* key, int64_t value)
{
    REQUIRE(h && key, "null argument");
    const int rc = set_switch(h, OPT_KEYS, key, value);
    if (rc <= 0) return rc;
    set_err("unknown option %s", key);
    return EXAADMM_ERR_INVALID;
}

extern "C" int exaadmm_get_option(exaadmm_handle* h, const char* key, int64_t* value)
{
    REQUIRE(h && key && value, "null argument");
    for (const OptKey& k : OPT_KEYS)
        if (strcmp(k.name, key) == 0) {
            *value = h->opt.*(k.field);
            return EXAADMM_OK;
        }
    set_err("unknown option %s", key);
    return EXAADMM_ERR_INVALID;
}

#if defined(EXA_TEST_HOOKS)
// Test build only (csrc/Makefile: libexaadmm_hip_hooks.so, -DEXA_TEST_HOOKS): fault injection for tests/test_gpu_resident.py.
//   "fail_resident_launch" n : the n-th resident launch of the handle reports a failed wait (0: off)
//   "flip_resident_ulp" 1    : the resident leg of the next self check gets one ulp added to one word of u
//   "p2p_delay_us" n         : every mailbox message of a partition's resident loop is delayed by n microseconds
//   "p2p_patient_ms" n       : the kernels in which a partition's ranks wait for each other between launches give up after n milliseconds instead of
//                              60 s (after exaadmm_dist_p2p_import; tests with thread ranks on ONE GPU, where a rank's kernel can be held up for good)
// ... and three switches that let a test reach code a product run selects by itself:
//   "resident_coop" 0        : plain instead of cooperative launches of the resident kernels (two ranks of a test sharing ONE GPU)
//   "resident_cap_limit" n   : treat the chip as holding at most n resident wavefronts (the two-tile form on small grids)
//   "heavy_select_small" 1   : the kernel that picks a launch's heavy wavefronts runs as one wavefront (ranks of a test sharing ONE GPU)
extern "C" int exaadmm_test_hook(exaadmm_handle* h, const char* key, int64_t value)
{
    REQUIRE(h && key, "null argument");
    const int rc = set_switch(h, TEST_KEYS, key, value);   // "resident_coop", "resident_cap_limit", "heavy_select_small"
    if (rc <= 0) return rc;
    if (strcmp(key, "fail_resident_launch") == 0) h->hook_fail_launch = (int)value;
    else if (strcmp(key, "flip_resident_ulp") == 0) h->hook_flip_ulp = (int)value;
    else if (strcmp(key, "p2p_delay_us") == 0) h->hook_p2p_delay_ticks = (int)(100 * value);
    else if (strcmp(key, "p2p_patient_ms") == 0) h->rpeers.patient_ticks = (long long)value * 100000ll;
    else {
        set_err("unknown test hook %s", key);
        return EXAADMM_ERR_INVALID;
    }
    return EXAADMM_OK;
}
#endif

// ------------------------------------------------------------------ device-side snapshot of the iterate
// exaadmm_state_save keeps a copy of everything an inner iteration reads and writes (u, v, z, z_prev, l, lz, the
// line-limit multipliers and penalty) in a second set of device buffers; exaadmm_state_restore brings it back.  bench.py
// times repeated K-step regions from ONE defined ADMM state with it (device-to-device copies, no PCIe).
extern "C" int exaadmm_state_save(exaadmm_handle* h)
{
    REQUIRE(h, "null handle");
    HIPCHK(hipSetDevice(h->device));
    int rc;
    const size_t nramp = (h->T > 1) ? (size_t)h->T * h->g1 : 0;
    if (h->snap.empty()) {
        for (int q = 0; q < 6; q++) {
            double* p = nullptr;
            if ((rc = dev_alloc(h, &p, (size_t)h->N))) return rc;
            h->snap.push_back(p);
        }
        for (int q = 0; q < 3; q++) {
            double* p = nullptr;
            if ((rc = dev_alloc(h, &p, (size_t)h->Lp))) return rc;
            h->snap.push_back(p);
        }
        for (int q = 0; q < 8 && nramp > 0; q++) {   // multi-period model: the ramp blocks (phat, s, z, z_prev, l, lz, multiplier, penalty)
            double* p = nullptr;
            if ((rc = dev_alloc(h, &p, nramp))) return rc;
            h->snap.push_back(p);
        }
    }
    double* src[17] = {h->d.u, h->d.v, h->d.z, h->d.zp, h->d.l, h->d.lz, h->d.ls1, h->d.ls2, h->d.mu,
                       h->d.r_u, h->d.r_s, h->d.r_z, h->d.r_zp, h->d.r_l, h->d.r_lz, h->d.r_mu, h->d.r_xi};
    for (size_t q = 0; q < h->snap.size(); q++)
        HIPCHK(hipMemcpyAsync(h->snap[q], src[q], (q < 6 ? (size_t)h->N : (q < 9 ? (size_t)h->Lp : nramp)) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    {
        const int* hist[4] = {h->nsets > 0 ? h->sets[0].st[2] : h->d.st_ticks, h->nsets > 1 ? h->sets[1].st[2] : nullptr, h->nsets > 2 ? h->sets[2].st[2] : nullptr, h->rsync.last3};
        for (int q = 0; q < 4; q++) {
            h->snap_hist_valid[q] = hist[q] != nullptr;
            if (!hist[q]) continue;
            if (!h->snap_hist[q] && (rc = dev_alloc(h, &h->snap_hist[q], (size_t)h->Lp))) return rc;
            HIPCHK(hipMemcpyAsync(h->snap_hist[q], hist[q], (size_t)h->Lp * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        }
        h->snap_resident_iters = h->resident_iters;
        h->snap_nsets = h->nsets;
        h->snap_last_inner_len = h->last_inner_len;
    }
    return sync_check(h, true);
}

extern "C" int exaadmm_state_restore(exaadmm_handle* h)
{
    REQUIRE(h && !h->snap.empty(), "exaadmm_state_save first");
    HIPCHK(hipSetDevice(h->device));
    const size_t nramp = (h->T > 1) ? (size_t)h->T * h->g1 : 0;
    double* dst[17] = {h->d.u, h->d.v, h->d.z, h->d.zp, h->d.l, h->d.lz, h->d.ls1, h->d.ls2, h->d.mu,
                       h->d.r_u, h->d.r_s, h->d.r_z, h->d.r_zp, h->d.r_l, h->d.r_lz, h->d.r_mu, h->d.r_xi};
    for (size_t q = 0; q < h->snap.size(); q++)
        HIPCHK(hipMemcpyAsync(dst[q], h->snap[q], (q < 6 ? (size_t)h->N : (q < 9 ? (size_t)h->Lp : nramp)) * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    {
        int* hist[4] = {h->nsets > 0 ? h->sets[0].st[2] : h->d.st_ticks, h->nsets > 1 ? h->sets[1].st[2] : nullptr, h->nsets > 2 ? h->sets[2].st[2] : nullptr, h->rsync.last3};
        for (int q = 0; q < 4 && h->nsets == h->snap_nsets; q++) {   // (the buffer sets were set up differently when the state was saved: the history stays)
            if (!hist[q]) continue;
            if (h->snap_hist_valid[q])
                HIPCHK(hipMemcpyAsync(hist[q], h->snap_hist[q], (size_t)h->Lp * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
            else   // (a buffer that did not exist when the state was saved held nothing then)
                HIPCHK(hipMemsetAsync(hist[q], 0, (size_t)h->Lp * sizeof(int), h->stream));
        }
        h->resident_iters = h->snap_resident_iters;
        h->last_inner_len = h->snap_last_inner_len;
    }
    h->residual_vectors_valid = false;
    return sync_check(h, true);
}

// ------------------------------------------------------------------ multi-GPU helpers
extern "C" int exaadmm_ctrl_begin(exaadmm_handle* h, int outer, int max_inner, double beta, int64_t nvar_global)
{
    REQUIRE(h && outer >= 1 && max_inner >= 1 && nvar_global >= 1, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    Ctrl* hc = h->h_ctrl;
    memset(hc, 0, sizeof(Ctrl));
    hc->max_inner = max_inner;
    hc->check_every = std::max(1, (int)h->par.check_every);
    hc->eps_pri = sqrt((double)nvar_global) / (2500.0 * outer);
    hc->beta = beta;
    HIPCHK(hipMemcpyAsync(h->d_ctrl, hc, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ctrl_mode = true;
    return EXAADMM_OK;
}

extern "C" int exaadmm_ctrl_step(exaadmm_handle* h, const double* dev_sums4)
{
    REQUIRE(h && dev_sums4 && h->ctrl_mode, "ctrl not armed");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ctrl_step, dim3(1), dim3(64), 0, h->stream, dev_sums4, h->d_ctrl);
    return sync_check(h);
}

extern "C" int exaadmm_ctrl_read(exaadmm_handle* h, int* done, int* inner, double res[4])
{
    REQUIRE(h && h->ctrl_mode, "ctrl not armed");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->h_ctrl, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    int rc = sync_check(h, true);
    if (rc) return rc;
    if (done) *done = h->h_ctrl->done;
    if (inner) *inner = h->h_ctrl->inner;
    if (res)
        for (int j = 0; j < 4; j++) res[j] = h->h_ctrl->res[j];
    return EXAADMM_OK;
}

extern "C" int exaadmm_ctrl_end(exaadmm_handle* h)
{
    REQUIRE(h, "null handle");
    h->ctrl_mode = false;
    return EXAADMM_OK;
}

extern "C" int exaadmm_set_async(exaadmm_handle* h, int async)
{
    REQUIRE(h, "null handle");
    h->async_mode = (async != 0);
    return EXAADMM_OK;
}

static VecTable vec_table(exaadmm_handle* h)
{
    VecTable t;
    for (int w = 0; w < 10; w++) t.p[w] = vec_ptr(h, w);
    return t;
}

extern "C" int exaadmm_set_halo_index(exaadmm_handle* h, int slot, const int32_t* which, const int64_t* idx, int64_t count)
{
    REQUIRE(h && slot >= 0 && slot < 8 && count >= 0, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    HaloSlot& s = h->halo[slot];
    if (s.d_idx) { (void)hipFree(s.d_idx); s.d_idx = nullptr; }
    if (s.d_which) { (void)hipFree(s.d_which); s.d_which = nullptr; }
    s.count = (int)count;
    if (count == 0) return EXAADMM_OK;
    REQUIRE(idx && which, "null index list");
    std::vector<int> t(count), w(count);
    for (int64_t i = 0; i < count; i++) {
        REQUIRE(idx[i] >= 0 && idx[i] < h->nvar, "halo index out of range (zero-based reference-layout index expected)");
        REQUIRE(which[i] >= 0 && which[i] < 10, "halo vector id out of range");
        t[i] = soa_index(h, idx[i]);
        w[i] = which[i];
    }
    HIPCHK(hipMalloc((void**)&s.d_idx, count * sizeof(int)));
    HIPCHK(hipMalloc((void**)&s.d_which, count * sizeof(int)));
    HIPCHK(hipMemcpy(s.d_idx, t.data(), count * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s.d_which, w.data(), count * sizeof(int), hipMemcpyHostToDevice));
    return EXAADMM_OK;
}

extern "C" int exaadmm_halo_pack(exaadmm_handle* h, int slot, double* dev_buf)
{
    REQUIRE(h && slot >= 0 && slot < 8, "bad argument");
    const HaloSlot& s = h->halo[slot];
    if (s.count == 0) return EXAADMM_OK;
    REQUIRE(dev_buf, "null buffer");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_gather, grid_for(s.count, 256), dim3(256), 0, h->stream, vec_table(h), s.d_which, s.d_idx, s.count, dev_buf);
    return sync_check(h);
}

extern "C" int exaadmm_halo_unpack(exaadmm_handle* h, int slot, const double* dev_buf)
{
    REQUIRE(h && slot >= 0 && slot < 8, "bad argument");
    const HaloSlot& s = h->halo[slot];
    if (s.count == 0) return EXAADMM_OK;
    REQUIRE(dev_buf, "null buffer");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_scatter, grid_for(s.count, 256), dim3(256), 0, h->stream, vec_table(h), s.d_which, s.d_idx, s.count, dev_buf);
    h->residual_vectors_valid = false;
    return sync_check(h);
}

// ------------------------------------------------------------------ distributed inner loop inside the library (RCCL)
#include "dist_native.h"
