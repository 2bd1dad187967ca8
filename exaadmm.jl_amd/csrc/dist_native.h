// Multi-GPU: one handle = one partition of the grid on one MI355X, one process per GPU, ONE exchange per inner
// iteration issued to RCCL from C++ on the handle's own stream.
// (Included at the end of exaadmm_hip.hip: uses the handle and the launch helpers defined there.)
//
// The reference has no multi-process path (acopf_model.jl:49-56 is vestigial scaffolding); the scheme is DESIGN.md
// section 6.  Buses are partitioned; a branch whose two buses live on different ranks (a cut branch) is held AND SOLVED
// by both: the two replicas see the same inputs and run the same code, so u, z, l, lz and the line-limit multipliers
// of a cut branch are bit-identical on both owners without ever travelling.  What a rank cannot compute itself is the
// half of such a branch's v that the OTHER rank's bus update produces: 4 doubles per cut branch and iteration, each way.
// Per inner iteration (the fused loop of exaadmm_hip.hip, unchanged apart from the exchange):
//     branch launch : z / l / residual terms of the previous iteration (prologue) + x-update, all local branches
//     bus launch    : bus update of the local buses; lanes of a cut branch's end also drop their 4 new v entries into
//                     the peer's message; the reduction block appends this rank's 4 residual sums to EVERY peer's message
//     exchange      : one grouped ncclSend / ncclRecv per peer (message = its cut branches' v entries + the 4 sums)
//     k_halo_unpack : scatter the arrived v entries; add the ranks' sums IN RANK ORDER; inner++, stop test
// No all-reduce and no separate control kernel: every rank adds the same numbers in the same order and therefore takes
// the same decision in the same iteration, one launch late like the single-GPU loop.  Round 2 needed nine launches, two
// exchanges and an all-reduce per iteration.
//
// RCCL is bound at run time (dlopen) so that the library keeps loading on hosts without it.  It must be the RCCL that
// sits on THE SAME HIP runtime as this library, because the handle's stream is handed to it: the copy next to the
// libamdhip64 this library is bound to is loaded by path; torch.distributed is only used by the caller to hand the
// 128-byte id around.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>
#include <mutex>

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // optional (reporting only): how many ranks the communicator really has, which RCCL it is
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

struct exaadmm_dist {
    RcclApi api;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    int64_t nvar_global = 0;
    // message to / from peer p: [4 doubles per cut branch shared with p, in the order both sides listed them][4 sums];
    // the two directions have the same layout, at msg_off[p] of the send / receive buffer
    std::vector<int64_t> ncut, msg_off;
    int n_cut = 0;
    double *send = nullptr, *recv = nullptr, *local = nullptr;
    int *d_pos = nullptr, *d_off = nullptr, *d_rcode = nullptr, *d_rpos = nullptr;
    // the exchange of an iteration runs on its own stream, beside the interior branch solves of the next iteration
    hipStream_t xstream = nullptr;
    std::vector<hipEvent_t> ev;        // fork / join and per-iteration hand-over events (also inside a captured batch)
    hipEvent_t xev[2] = {nullptr, nullptr};   // around one sampled exchange (group + unpack) per directly enqueued batch
    bool xev_armed = false, xev_recorded = false;   // armed: the next directly enqueued batch brackets one exchange; recorded: read them after a synchronisation
    double exchange_us_sum = 0.0;
    int64_t exchange_us_n = 0;
    // a batch of iterations INCLUDING its exchanges as one hipGraph: offered when the bound RCCL captured, instantiated and
    // replayed a sums exchange with every peer at connect time (dist_probe); otherwise every batch is enqueued directly
    bool graph_ok = false;
    // mailboxes of the resident loop (kernels.h ResidentPeers): ours (fine-grained device memory the peers write into over xGMI) and
    // the peers', mapped here -- by IPC handle across processes, by address between handles of one process
    double* box = nullptr;
    size_t box_entries = 0;
    int *d_cut_code = nullptr;
    unsigned long long* d_twin = nullptr;
    unsigned xcount = 0;       // host-issued sums exchanges through the mailboxes so far (the same on every rank)
    int* h_xfail = nullptr;    // a host-issued exchange that found no peer leaves 1 here: pinned HOST memory the kernel writes to (d_xfail: its device
    int* d_xfail = nullptr;    //  address), read by the host at the call's synchronisation without a copy of its own (dist_check_exchange)
    std::vector<void*> ipc_opened;
    int64_t graph_batches = 0, direct_batches = 0;
    int probe_result = -1;   // -1 not run, 0 capture refused (direct enqueue), 1 captured exchange replayed correctly
};

static RcclApi g_rccl;   // process-wide function table
static std::string g_rccl_path;
static std::string g_rccl_override;   // exaadmm_dist_set_rccl_library: tried first
static std::mutex g_rccl_mutex;   // handles may be attached from different host threads

// Which communication library to bind instead of the librccl next to the HIP runtime (a site's own build; the tests' in-process
// stand-in).  Process-wide and only before the first bind: the function table is shared by every handle.
extern "C" int exaadmm_dist_set_rccl_library(const char* path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    const std::string want = path ? path : "";
    if (g_rccl.lib && want != g_rccl_override) {
        set_err("a communication library is already bound (%s): set the override before the first attach", g_rccl_path.c_str());
        return EXAADMM_ERR_INVALID;
    }
    g_rccl_override = want;
    return EXAADMM_OK;
}

// which RCCL the library bound (empty before the first exaadmm_dist_unique_id / multi-rank exaadmm_dist_prepare)
extern "C" const char* exaadmm_dist_rccl_path(void) { return g_rccl_path.c_str(); }

static int rccl_bind(RcclApi* out)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.lib) {
        std::vector<std::string> names;
        if (!g_rccl_override.empty()) names.push_back(g_rccl_override);
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {   // directory of the HIP runtime in use
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        void* lib = nullptr;
        std::string path;
        for (const std::string& n : names)
            if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) {
                path = n;
                break;
            }
        if (!lib) {
            set_err("RCCL not found next to the HIP runtime (exaadmm_dist_set_rccl_library names another one): %s", dlerror());
            return EXAADMM_ERR_INVALID;
        }
        RcclApi b;
        const char* missing = nullptr;
#define EXA_SYM(field, name)                              \
    do {                                                  \
        *(void**)(&b.field) = dlsym(lib, name);           \
        if (!b.field && !missing) missing = name;         \
    } while (0)
        EXA_SYM(GetUniqueId, "ncclGetUniqueId");
        EXA_SYM(CommInitRank, "ncclCommInitRank");
        EXA_SYM(CommDestroy, "ncclCommDestroy");
        EXA_SYM(GroupStart, "ncclGroupStart");
        EXA_SYM(GroupEnd, "ncclGroupEnd");
        EXA_SYM(Send, "ncclSend");
        EXA_SYM(Recv, "ncclRecv");
        EXA_SYM(GetErrorString, "ncclGetErrorString");
#undef EXA_SYM
        *(void**)(&b.CommCount) = dlsym(lib, "ncclCommCount");
        *(void**)(&b.GetVersion) = dlsym(lib, "ncclGetVersion");
        if (missing) {   // a rejected library leaves nothing behind: no handle, no half-filled table, no path
            set_err("RCCL symbol %s missing in %s", missing, path.c_str());
            (void)dlclose(lib);
            return EXAADMM_ERR_INVALID;
        }
        b.lib = lib;
        g_rccl = b;
        g_rccl_path = path;
    }
    *out = g_rccl;
    return 0;
}

#define NCCLCHK(d, expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess) {                                                                          \
            set_err("%s failed: %s (%s:%d)", #expr, (d)->api.GetErrorString(r__), __FILE__, __LINE__);     \
            return EXAADMM_ERR_HIP;                                                                        \
        }                                                                                                  \
    } while (0)

static int dist_probe(exaadmm_handle* h);
static bool dist_has_peers(const exaadmm_handle* h) { return h->dist != nullptr && h->dist->nranks > 1; }
static bool dist_graph_allowed(const exaadmm_handle* h) { return h->dist != nullptr && h->dist->graph_ok; }
// After the synchronisation of a call that issued a sums exchange through the mailboxes (k_p2p_exchange4): did one of them give up on a peer?  The
// sums it delivered are NaN and the loop control it fed is stopped; the call fails -- loudly: a caller that went on would compare against NaN.
static int dist_check_exchange(exaadmm_handle* h)
{
    if (h->dist == nullptr || h->dist->h_xfail == nullptr || *static_cast<volatile int*>(h->dist->h_xfail) == 0) return 0;
    set_err("rank %d of %d: a peer did not answer a sums exchange through the mailboxes within %.0f s; the results of this call are not valid", h->dist->rank,
            h->dist->nranks, (double)h->rpeers.patient_ticks / 1e8);
    return EXAADMM_ERR_HIP;
}
static void dist_graph_refused(exaadmm_handle* h)
{
    if (h->dist) h->dist->graph_ok = false;
}
static thread_local int64_t g_exchanges = 0;
static void dist_count_batch(exaadmm_handle* h, bool graph, int exchanges)
{
    if (!h->dist) return;
    (graph ? h->dist->graph_batches : h->dist->direct_batches) += 1;
    g_exchanges += exchanges;   // (a replayed batch enqueues its exchanges without passing through dist_exchange)
}
// arm = true: the next directly enqueued batch brackets ONE of its exchanges (group + unpack) with HIP events on the exchange
// stream; arm = false (after the batch's synchronisation): read them
static void dist_sample_exchange(exaadmm_handle* h, bool arm)
{
    exaadmm_dist* d = h->dist;
    if (!d || !d->xev[0]) return;
    if (arm) {
        d->xev_armed = true;
        return;
    }
    d->xev_armed = false;
    if (!d->xev_recorded) return;
    d->xev_recorded = false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, d->xev[0], d->xev[1]) == hipSuccess && ms > 0.f) {
        d->exchange_us_sum += 1e3 * (double)ms;
        d->exchange_us_n += 1;
    } else
        (void)hipGetLastError();
}
static int64_t nvar_for_tolerances(const exaadmm_handle* h) { return h->dist ? h->dist->nvar_global : h->nvar_ctrl; }

extern "C" int exaadmm_dist_unique_id(void* out, int64_t nbytes)
{
    REQUIRE(out && nbytes == (int64_t)sizeof(ncclUniqueId), "buffer must hold 128 bytes (ncclUniqueId)");
    exaadmm_dist tmp;
    int rc = rccl_bind(&tmp.api);
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(&tmp, tmp.api.GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return EXAADMM_OK;
}

extern "C" int exaadmm_dist_detach(exaadmm_handle* h)
{
    REQUIRE(h, "null handle");
    exaadmm_dist* d = h->dist;
    if (!d) return EXAADMM_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (d->xstream) (void)hipStreamSynchronize(d->xstream);
    if (d->comm) (void)d->api.CommDestroy(d->comm);
    for (hipEvent_t e : d->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->xev)
        if (e) (void)hipEventDestroy(e);
    if (d->xstream) (void)hipStreamDestroy(d->xstream);
    void* bufs[] = {d->send, d->recv, d->local, d->d_pos, d->d_off, d->d_rcode, d->d_rpos, d->box, d->d_cut_code, d->d_twin};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (d->h_xfail) (void)hipHostFree(d->h_xfail);
    for (void* q : d->ipc_opened) (void)hipIpcCloseMemHandle(q);
    h->p2p_ready = false;
    h->rpeers = ResidentPeers{};
    delete d;
    h->dist = nullptr;
    Dev& dv = h->d;
    dv.nranks = 1;
    dv.rank = 0;
    dv.halo_pos = dv.halo_off = nullptr;
    dv.halo_send = dv.halo_local = nullptr;
    dv.halo_recv = nullptr;
    h->dev_dirty = true;
    invalidate_graph(h);
    return EXAADMM_OK;
}

static int dist_prepare_impl(exaadmm_handle* h, int nranks, int rank, int64_t nvar_global, const int64_t* ncut_per_peer,
                             const int64_t* cut_branch)
{
    REQUIRE(h && nranks >= 1 && rank >= 0 && rank < nranks && nvar_global >= 1, "bad argument");
    REQUIRE(h->T == 1, "a multi-period handle cannot be a partition");
    REQUIRE(nranks == 1 || ncut_per_peer, "null cut list");
    HIPCHK(hipSetDevice(h->device));
    int rc = exaadmm_dist_detach(h);
    if (rc) return rc;
    exaadmm_dist* d = new exaadmm_dist();
    h->dist = d;
    d->nranks = nranks;
    d->rank = rank;
    d->nvar_global = nvar_global;
    d->ncut.assign(nranks, 0);
    d->msg_off.assign(nranks, 0);
    int64_t total = 0, off = 0;
    for (int p = 0; p < nranks; p++) {
        const int64_t n = (nranks > 1) ? ncut_per_peer[p] : 0;
        REQUIRE(n >= 0, "negative cut count");
        REQUIRE(p != rank || n == 0, "a rank shares no branch with itself");
        d->ncut[p] = n;
        d->msg_off[p] = off;
        if (p != rank) off += 4 * n + 4;
        total += n;
    }
    d->n_cut = (int)total;
    if (nranks == 1) return EXAADMM_OK;   // no peers: the fused loop runs as on a single GPU, nothing to exchange
    REQUIRE(total == 0 || cut_branch, "null cut list");
    // every branch with exactly one local end must be listed once, and nothing else (exaadmm_create flagged them)
    std::vector<int> pos(h->Lp, -1), rcode((size_t)std::max<int64_t>(total, 1)), rpos((size_t)std::max<int64_t>(total, 1));
    std::vector<int> offs(2 * (size_t)nranks, -1);
    int64_t expected = 0;
    for (int b = 0; b < h->L; b++) expected += (h->cut_side[b] != 0);
    REQUIRE(expected == total, "the cut lists must name exactly the branches with one end on this partition's buses");
    int64_t i = 0;
    for (int p = 0; p < nranks; p++) {
        for (int64_t k = 0; k < d->ncut[p]; k++, i++) {
            const int64_t b = cut_branch[i];
            REQUIRE(b >= 0 && b < h->L && h->cut_side[b] != 0, "cut list names a branch that is not cut");
            REQUIRE(pos[h->perm[b]] < 0, "cut list names a branch twice");
            pos[h->perm[b]] = (int)(d->msg_off[p] + 4 * k);
            // the half that ARRIVES is the one on the peer's bus
            rcode[(size_t)i] = ((h->cut_side[b] == 1 ? END_TO : END_FROM) << 28) | h->perm[b];
            rpos[(size_t)i] = (int)(d->msg_off[p] + 4 * k);
        }
        if (p != rank) offs[p] = offs[(size_t)nranks + p] = (int)(d->msg_off[p] + 4 * d->ncut[p]);
    }
    if ((rc = rccl_bind(&d->api))) return rc;
    HIPCHK(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
    for (auto& e : d->xev) HIPCHK(hipEventCreate(&e));
    const size_t nbuf = (size_t)std::max<int64_t>(off, 4);
    HIPCHK(hipMalloc((void**)&d->send, nbuf * sizeof(double)));
    HIPCHK(hipMalloc((void**)&d->recv, nbuf * sizeof(double)));
    HIPCHK(hipMalloc((void**)&d->local, 4 * sizeof(double)));
    HIPCHK(hipMemset(d->send, 0, nbuf * sizeof(double)));
    HIPCHK(hipMemset(d->recv, 0, nbuf * sizeof(double)));
    HIPCHK(hipMemset(d->local, 0, 4 * sizeof(double)));
    HIPCHK(hipMalloc((void**)&d->d_pos, pos.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&d->d_off, offs.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&d->d_rcode, rcode.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&d->d_rpos, rpos.size() * sizeof(int)));
    HIPCHK(hipMemcpy(d->d_pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->d_off, offs.data(), offs.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->d_rcode, rcode.data(), rcode.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->d_rpos, rpos.data(), rpos.size() * sizeof(int), hipMemcpyHostToDevice));
    if (nranks <= P2P_MAX_RANKS) {
        // the mailbox of the resident loop: one 128-byte entry per cut branch (in the order of the cut lists), then one per rank for the
        // residual sums and one per rank for the end-of-launch status.  Fine-grained device memory: stores from a peer GPU become
        // visible to a running kernel here.  cut_code[storage slot] = (entry << 1) | (1: the TO end's bus is the peer's)
        d->box_entries = (size_t)total + 3 * (size_t)nranks;   // cut branches | residual sums | end-of-launch status | host-issued sums exchanges
        void* bx = nullptr;
        hipError_t e = hipExtMallocWithFlags(&bx, d->box_entries * P2P_ENTRY * sizeof(double), hipDeviceMallocFinegrained);
        if (e != hipSuccess) {   // (no fine-grained memory on this system: the partition keeps the two-launch loop)
            (void)hipGetLastError();
            bx = nullptr;
            d->box_entries = 0;
        }
        d->box = static_cast<double*>(bx);
        if (d->box) {
            HIPCHK(hipMemset(d->box, 0, d->box_entries * P2P_ENTRY * sizeof(double)));
            std::vector<int> code(h->Lp, -1);
            for (int64_t c = 0; c < total; c++) {
                const int slot = rcode[(size_t)c] & END_INDEX;
                const bool remote_is_to = ((rcode[(size_t)c] >> 28) & 3) == END_TO;   // (rcode names the half that ARRIVES: the end on the peer's bus)
                code[slot] = (int)(c << 1) | (remote_is_to ? 1 : 0);
            }
            HIPCHK(hipMalloc((void**)&d->d_cut_code, code.size() * sizeof(int)));
            HIPCHK(hipMemcpy(d->d_cut_code, code.data(), code.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(hipMalloc((void**)&d->d_twin, (size_t)(total + nranks) * sizeof(unsigned long long)));
            HIPCHK(hipHostMalloc((void**)&d->h_xfail, 64, hipHostMallocMapped));
            *d->h_xfail = 0;
            HIPCHK(hipHostGetDevicePointer((void**)&d->d_xfail, d->h_xfail, 0));
        }
    }
    Dev& dv = h->d;
    dv.nranks = nranks;
    dv.rank = rank;
    dv.halo_pos = d->d_pos;
    dv.halo_off = d->d_off;
    dv.halo_send = d->send;
    dv.halo_recv = d->recv;
    dv.halo_local = d->local;
    h->dev_dirty = true;
    invalidate_graph(h);
    return EXAADMM_OK;
}

// Local half of attaching a partition to its peers: binds RCCL, checks the cut lists against the grid, allocates the
// message buffers.  Nothing here talks to another rank, so a failure is this rank's alone and the caller can agree on it
// with its peers BEFORE anybody enters the communicator set-up (exaadmm_dist_connect), which would otherwise wait forever
// for the rank that failed.  cut_branch: for every peer in rank order, the zero-based local ids of the branches shared
// with it, in the order both sides agreed on (ascending global branch id).
extern "C" int exaadmm_dist_prepare(exaadmm_handle* h, int nranks, int rank, int64_t nvar_global, const int64_t* ncut_per_peer,
                                    const int64_t* cut_branch)
{
    const int rc = dist_prepare_impl(h, nranks, rank, nvar_global, ncut_per_peer, cut_branch);
    if (rc != EXAADMM_OK && h != nullptr && h->dist != nullptr) {   // no half-attached handle survives a failure
        const std::string keep = g_err;
        (void)exaadmm_dist_detach(h);
        g_err = keep;
    }
    return rc;
}

// Collective half: ncclCommInitRank over the ranks that all passed exaadmm_dist_prepare.
extern "C" int exaadmm_dist_connect(exaadmm_handle* h, const void* unique_id, int64_t id_bytes)
{
    REQUIRE(h && h->dist, "exaadmm_dist_prepare first");
    exaadmm_dist* d = h->dist;
    REQUIRE(unique_id && id_bytes == (int64_t)sizeof(ncclUniqueId), "unique id must be the 128 bytes of exaadmm_dist_unique_id");
    REQUIRE(d->comm == nullptr, "already connected");
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if (!d->api.lib && (rc = rccl_bind(&d->api))) return rc;   // (one-rank communicator of the self test)
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = d->api.CommInitRank(&d->comm, d->nranks, id, d->rank);
    if (r != ncclSuccess) {
        set_err("ncclCommInitRank failed: %s", d->api.GetErrorString(r));
        d->comm = nullptr;
        const std::string keep = g_err;
        (void)exaadmm_dist_detach(h);
        g_err = keep;
        return EXAADMM_ERR_HIP;
    }
    rc = dist_probe(h);   // (collective as well: every rank that got its communicator runs it)
    if (rc != EXAADMM_OK) {
        const std::string keep = g_err;
        (void)exaadmm_dist_detach(h);
        g_err = keep;
    }
    return rc;
}

// what the distributed loop of this handle did so far (bench.py's N > 1 line): ranks of the RCCL communicator as RCCL counts
// them (-1: the bound library does not say), RCCL version code (0: unknown), mean duration in microseconds of the sampled
// exchanges (group + unpack kernel, HIP events on the exchange stream; 0 when nothing was sampled), batches replayed from a
// hipGraph / enqueued directly, result of the capture probe (-1 not run, 0 refused, 1 ok)
extern "C" int exaadmm_dist_stats(exaadmm_handle* h, int* comm_ranks, int* rccl_version, double* exchange_us, int64_t* graph_batches,
                                  int64_t* direct_batches, int* probe)
{
    REQUIRE(h && h->dist, "not a partition");
    exaadmm_dist* d = h->dist;
    int n = -1, v = 0;
    if (d->comm && d->api.CommCount && d->api.CommCount(d->comm, &n) != ncclSuccess) n = -1;
    if (d->api.GetVersion && d->api.GetVersion(&v) != ncclSuccess) v = 0;
    if (comm_ranks) *comm_ranks = d->comm ? n : (d->nranks == 1 ? 1 : -1);
    if (rccl_version) *rccl_version = v;
    if (exchange_us) *exchange_us = d->exchange_us_n > 0 ? d->exchange_us_sum / (double)d->exchange_us_n : 0.0;
    if (graph_batches) *graph_batches = d->graph_batches;
    if (direct_batches) *direct_batches = d->direct_batches;
    if (probe) *probe = d->probe_result;
    return EXAADMM_OK;
}

// prepare + connect in one call, for callers that have no way (or need) to agree on local failures first
extern "C" int exaadmm_dist_attach(exaadmm_handle* h, int nranks, int rank, const void* unique_id, int64_t id_bytes,
                                   int64_t nvar_global, const int64_t* ncut_per_peer, const int64_t* cut_branch)
{
    int rc = exaadmm_dist_prepare(h, nranks, rank, nvar_global, ncut_per_peer, cut_branch);
    if (rc) return rc;
    if (nranks > 1 || unique_id != nullptr) rc = exaadmm_dist_connect(h, unique_id, id_bytes);   // (one rank: optional)
    return rc;
}

// ---- mailboxes of the resident loop: who can write where
// exaadmm_dist_p2p_export: 256 bytes that describe this rank's mailbox -- process id, device, address, IPC handle, its cut counts per
// peer; the caller hands every rank's blob to every rank (torch.distributed.all_gather, MPI; the thread-rank tests: a shared list);
// exaadmm_dist_p2p_import(all blobs in rank order): maps every peer's mailbox (same process: by address; another process:
// hipIpcOpenMemHandle + peer access) and computes where each local cut branch's twin lane and each peer's sums entry for this rank
// live.  From then on exaadmm_run_inner / _run_fixed / _solve_two_level use the resident loop on this partition when its wavefronts
// fit the chip (exaadmm_loop_kind says 2).  Local: nothing here talks to a peer; callers agree on the outcome before they rely on it.
struct P2PBlob {
    unsigned long long magic;
    long long pid;
    int device, nranks, rank, has_box;
    unsigned long long address;
    hipIpcMemHandle_t ipc;
    int ncut[P2P_MAX_RANKS];
    // which resident form this rank's partition would run with its peers' mailboxes mapped (resident_mode_if: 0 none, 1 one tile, 2 two tiles per
    // wavefront): EVERY rank must run a resident form or none may -- a rank on the two-launch loop would leave its peers' launches in their capped
    // waits -- and only when all of them run the one-tile form may the first launch of an inner loop be split for the heavy wavefronts (launch
    // lengths must agree across ranks).  Every rank reads the same blobs, so every rank reaches the same verdict without another exchange.
    int resident_form;
};
static_assert(sizeof(P2PBlob) <= 256, "blob size is part of the ABI (EXAADMM_P2P_BLOB_BYTES)");
constexpr unsigned long long P2P_MAGIC = 0x6578616164326d70ull;

extern "C" int exaadmm_dist_p2p_export(exaadmm_handle* h, void* blob, int64_t nbytes)
{
    REQUIRE(h && h->dist && blob && nbytes >= 256, "needs a prepared partition and 256 bytes");
    exaadmm_dist* d = h->dist;
    HIPCHK(hipSetDevice(h->device));
    P2PBlob b;
    memset(&b, 0, sizeof(b));
    b.magic = P2P_MAGIC;
    b.pid = (long long)getpid();
    b.device = h->device;
    b.nranks = d->nranks;
    b.rank = d->rank;
    b.has_box = (d->box != nullptr && d->nranks <= P2P_MAX_RANKS) ? 1 : 0;
    b.address = (unsigned long long)(uintptr_t)d->box;
    if (b.has_box) {
        if (hipIpcGetMemHandle(&b.ipc, d->box) != hipSuccess) {
            (void)hipGetLastError();
            memset(&b.ipc, 0, sizeof(b.ipc));   // (peers in this process use the address; peers in other processes will find no usable handle)
        }
        for (int p = 0; p < d->nranks; p++) b.ncut[p] = (int)d->ncut[p];
    }
    b.resident_form = b.has_box ? resident_mode_if(h, true) : 0;
    memset(blob, 0, 256);
    memcpy(blob, &b, sizeof(b));
    return EXAADMM_OK;
}

extern "C" int exaadmm_dist_p2p_import(exaadmm_handle* h, const void* blobs, int64_t nbytes_each)
{
    REQUIRE(h && h->dist && blobs && nbytes_each >= 256, "needs a prepared partition and every rank's 256-byte blob");
    exaadmm_dist* d = h->dist;
    h->p2p_ready = false;
    REQUIRE(d->nranks > 1 && d->nranks <= P2P_MAX_RANKS, "mailboxes serve 2 .. 16 ranks");
    REQUIRE(d->box != nullptr, "this rank has no mailbox (no fine-grained device memory)");
    HIPCHK(hipSetDevice(h->device));
    std::vector<P2PBlob> bl((size_t)d->nranks);
    for (int p = 0; p < d->nranks; p++) {
        memcpy(&bl[p], static_cast<const char*>(blobs) + (size_t)p * nbytes_each, sizeof(P2PBlob));
        REQUIRE(bl[p].magic == P2P_MAGIC && bl[p].rank == p && bl[p].nranks == d->nranks, "blob of the wrong rank / job");
        REQUIRE(bl[p].has_box == 1, "a peer has no mailbox");
    }
    bool all_one_tile = true;
    for (int p = 0; p < d->nranks; p++) {
        // (same blobs on every rank -> the same refusal on every rank: nobody is left with a resident loop its peers do not run)
        REQUIRE(bl[p].resident_form != 0, "a rank's partition does not fit the resident loop (size, options or stream): all ranks keep two launches per iteration");
        all_one_tile = all_one_tile && bl[p].resident_form == 1;
    }
    h->peers_one_tile = all_one_tile;
    REQUIRE(bl[d->rank].address == (unsigned long long)(uintptr_t)d->box, "our own blob does not describe this handle");
    std::vector<unsigned long long> base((size_t)d->nranks, 0ull);
    for (int p = 0; p < d->nranks; p++) {
        if (p == d->rank) continue;
        REQUIRE(bl[p].ncut[d->rank] == (int)d->ncut[p], "the two sides of a cut disagree about its size");
        if (bl[p].pid == (long long)getpid()) {
            base[p] = bl[p].address;   // a handle of this process (thread ranks; several GPUs driven by one process)
            if (bl[p].device != h->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(bl[p].device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    set_err("hipDeviceEnablePeerAccess(%d) failed: %s", bl[p].device, hipGetErrorString(e));
                    return EXAADMM_ERR_HIP;
                }
                (void)hipGetLastError();
            }
        } else {
            void* q = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&q, bl[p].ipc, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_err("hipIpcOpenMemHandle for rank %d's mailbox failed: %s", p, hipGetErrorString(e));
                return EXAADMM_ERR_HIP;
            }
            d->ipc_opened.push_back(q);
            base[p] = (unsigned long long)(uintptr_t)q;
        }
    }
    // entry of cut (p, k) on rank p: its cut list is peer-major in rank order as well -- (cuts with ranks < us) + k
    // (twin addresses [0, n_cut), then this rank's sums entry in every peer's mailbox [n_cut, n_cut + nranks): one device table)
    std::vector<unsigned long long> twin((size_t)d->n_cut + (size_t)d->nranks, 0ull);
    ResidentPeers rp = {};
    int64_t c = 0;
    for (int p = 0; p < d->nranks; p++) {
        int64_t before = 0, total_p = 0;
        for (int q = 0; q < d->nranks; q++) {
            if (q < d->rank) before += bl[p].ncut[q];
            total_p += bl[p].ncut[q];
        }
        for (int64_t k = 0; k < d->ncut[p]; k++, c++) twin[(size_t)c] = base[p] + (unsigned long long)(before + k) * P2P_ENTRY * sizeof(double);
        twin[(size_t)d->n_cut + p] = (p == d->rank) ? 0ull : base[p] + (unsigned long long)(total_p + d->rank) * P2P_ENTRY * sizeof(double);
    }
    HIPCHK(hipMemcpy(d->d_twin, twin.data(), twin.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    rp.cut_code = d->d_cut_code;
    rp.twin = d->d_twin;
    rp.sums_at = d->d_twin + d->n_cut;
    rp.box = d->box;
    rp.nranks = d->nranks;
    rp.rank = d->rank;
    rp.ncut = d->n_cut;
    rp.epoch = 0;
    rp.patient_ticks = P2P_PATIENT_TICKS;
    h->rpeers = rp;
    h->p2p_epoch = 0;
    h->p2p_ready = true;
    invalidate_graph(h);
    return EXAADMM_OK;
}

#if defined(EXA_PROFILE_PHASES) || defined(EXA_TEST_HOOKS)
// diagnostics (scripts/p2p_debug.py; diagnostic and test builds only): this rank's mailbox as it is now, 16 doubles per entry
extern "C" int exaadmm_debug_p2p_box(exaadmm_handle* h, double* out, int64_t n_doubles)
{
    REQUIRE(h && h->dist && h->dist->box && out, "no mailbox");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = std::min<size_t>((size_t)n_doubles, h->dist->box_entries * P2P_ENTRY);
    HIPCHK(hipMemcpy(out, h->dist->box, n * sizeof(double), hipMemcpyDeviceToHost));
    return (int)(h->dist->box_entries);
}

// diagnostics: when the last resident launch of a partition with peers started and ended on the GPU (constant 100 MHz clock)
extern "C" int exaadmm_debug_resident_window(exaadmm_handle* h, unsigned long long* out2)
{
    REQUIRE(h && h->rsync.fail && out2, "no resident launch yet");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(out2, h->rsync.fail + 1, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int w[4];
    HIPCHK(hipMemcpy(w, h->rsync.contrib, sizeof(w), hipMemcpyDeviceToHost));   // contrib, decided, stop, fail
    out2[2] = ((unsigned long long)(unsigned)w[0] << 32) | (unsigned)w[1];
    out2[3] = ((unsigned long long)(unsigned)w[2] << 32) | (unsigned)w[3];
    return EXAADMM_OK;
}
#endif

// number of grouped exchanges this handle has enqueued since it was attached (tests: exactly one per inner iteration)
extern "C" int64_t exaadmm_dist_exchange_count(void) { return g_exchanges; }

// The one exchange: per peer one send and one receive of (its cut branches' v entries +) the four sums, grouped; then
// the unpack kernel, both on `stream`.  with_v = false: the sums only (end of an inner loop, ||z|| of the outer loop).
// sums_mode: k_halo_unpack (0 = the message carries sums, 1 = if ctrl->pending says so, 2 = it carries none).
static int dist_exchange(exaadmm_handle* h, const Dev& dv, bool with_v, Ctrl* c, double* out, int sums_mode, hipStream_t stream)
{
    exaadmm_dist* d = h->dist;
    if (!with_v && h->p2p_ready && h->opt.dist_p2p != 0 && h->resident_downgrades == 0) {
        // sums only, and the peers' mailboxes are mapped: one wavefront instead of pack kernel + grouped send / receive + unpack kernel
        // (k_halo_pack_sums has just filled halo_local with this rank's four sums -- read from there)
        d->xcount += 1;
        hipLaunchKernelGGL(k_p2p_exchange4, dim3(1), dim3(64), 0, stream, h->rpeers, d->xcount, (const double*)dv.halo_local, c, out, sums_mode, d->d_xfail);
        HIPCHK(hipGetLastError());
        return 0;
    }
    REQUIRE(d->comm != nullptr, "partition has peers but no communicator (exaadmm_dist_connect)");
    NCCLCHK(d, d->api.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int p = 0; p < d->nranks && bad == ncclSuccess; p++) {
        if (p == d->rank) continue;
        const size_t off = (size_t)(with_v ? d->msg_off[p] : d->msg_off[p] + 4 * d->ncut[p]);
        const size_t n = (size_t)(with_v ? 4 * d->ncut[p] + 4 : 4);
        bad = d->api.Send(d->send + off, n, ncclDouble, p, d->comm, stream);
        if (bad == ncclSuccess) bad = d->api.Recv(d->recv + off, n, ncclDouble, p, d->comm, stream);
    }
    const ncclResult_t ge = d->api.GroupEnd();   // always closed, also after a failed call inside the group
    if (bad != ncclSuccess || ge != ncclSuccess) {
        set_err("halo exchange failed: %s", d->api.GetErrorString(bad != ncclSuccess ? bad : ge));
        return EXAADMM_ERR_HIP;
    }
    {   // (counted when it is enqueued to run: a captured exchange runs -- and is counted, dist_count_batch -- with every replay of its graph)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs == hipStreamCaptureStatusNone) g_exchanges++;
        (void)hipGetLastError();
    }
    const int nc = with_v ? d->n_cut : 0;
    hipLaunchKernelGGL(k_halo_unpack, grid_for(std::max(nc, 1), 256), dim3(256), 0, stream, dv, d->d_rcode, d->d_rpos, nc, c, out, sums_mode);
    return 0;
}

// Iterations [j0, j0 + n) of the fused loop on a partition with peers (j = position relative to ctrl->base; launch j uses the
// buffer set j mod 3).  What an iteration consists of is what it consists of on one GPU -- branch launch (z / l pass of the
// previous iteration in its prologue), bus launch -- plus ONE exchange; what changes is WHEN things may start.  Only a cut
// branch's next solve needs the halo (the bus update never does), so the branch launch is split (exaadmm_create stores the
// cut branches last):
//     stream M :  ... Bus_j | Int_{j+1} ....................................| Bus_{j+1} | Int_{j+2} ...
//     stream X :            | exchange_j | unpack_j | Cut_{j+1} |                      | exchange_{j+1} ...
// Int = the blocks in front of the first cut branch + the generator blocks, Cut = the rest.  The iteration costs
// max(interior solves, exchange + unpack + cut solves) + bus launch instead of their sum.  The stop test rides in unpack_j (it
// needs the peers' sums of the z / l pass that launch j made for iteration j - 1), so Int_{j+1} runs TWO launches ahead of the
// decision about iteration j - 1: three buffer sets (spec_prepare) keep that iterate intact, and every kernel that starts after
// the decision returns at once (ctrl->done).  The host enqueues in the order of a software pipeline (Int_{j+1} BEFORE
// exchange_j), so that the overlap also exists under a host-blocking communication library (the tests' stand-in).
static int enqueue_dist_iterations(exaadmm_handle* h, Ctrl* c, int j0, int n, bool first_of_loop, hipEvent_t* bev)
{
    exaadmm_dist* d = h->dist;
    // option "dist_overlap" = 0 (A/B switch): everything on ONE stream, in the same order -- same results, the exchange fully exposed
    const bool overlap = h->opt.dist_overlap != 0;
    hipStream_t M = h->stream, X = overlap ? d->xstream : h->stream;
    const int nb = branch_blocks(h), ng = gen_blocks(h), nbi = interior_blocks(h), nbus = (int)grid_for(h->B, BUS_BLOCK / BUS_W).x;
    while ((int)d->ev.size() < 2 * n + 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->ev.push_back(e);
    }
    // (bev: exaadmm_run_fixed's ms_branch -- the INTERIOR launch only; the cut launch runs beside it on the exchange stream.  The pair is
    //  recorded for every iteration, also when a partition has no whole interior block and no generators: an empty interval)
    auto launch_int = [&](int j) {
        if (bev) (void)hipEventRecord(bev[2 * (j - j0)], M);
        if (nbi + ng > 0)
            hipLaunchKernelGGL((k_branch_t<false, true>), dim3(nbi + ng), dim3(BR_BLOCK), 0, M, h->d_dspec[j % 3][1], h->bo, h->par.scale, 0, c, ng, j, h->baseMVA);
        if (bev) (void)hipEventRecord(bev[2 * (j - j0) + 1], M);
    };
    auto launch_cut = [&](int j) {
        if (nb - nbi <= 0) return;
        hipLaunchKernelGGL((k_branch_t<false, true>), dim3(nb - nbi), dim3(BR_BLOCK), 0, X, h->d_dspec[j % 3][2], h->bo, h->par.scale, 0, c, 0, j, h->baseMVA);
    };
    // fork: X joins behind whatever M has been given so far (state uploads, the previous batch)
    if (overlap) {
        HIPCHK(hipEventRecord(d->ev[0], M));
        HIPCHK(hipStreamWaitEvent(X, d->ev[0], 0));
    }
    launch_int(j0);
    launch_cut(j0);
    int rc = 0;
    for (int j = j0; j < j0 + n; j++) {
        const Dev& dv = h->dspec[j % 3][0];
        const bool has_sums = !(first_of_loop && j == j0);   // (the first launch of a loop has no z / l pass in its prologue)
        hipEvent_t e_cut = d->ev[2 + 2 * (j - j0)], e_bus = d->ev[3 + 2 * (j - j0)];
        if (overlap) {
            HIPCHK(hipEventRecord(e_cut, X));
            HIPCHK(hipStreamWaitEvent(M, e_cut, 0));
        }
        Ctrl* red = has_sums ? c : nullptr;
        if (dv.rec != nullptr)
            hipLaunchKernelGGL((k_bus_group_t<true>), dim3(nbus + (red ? 1 : 0)), dim3(BUS_BLOCK), 0, M, (const Ctrl*)c, red, (const double*)dv.rec, dv.EndCount, dv.SlotCode, dv.busc, dv.B, dv, h->baseMVA);
        else
            hipLaunchKernelGGL((k_bus_group_t<false>), dim3(nbus + (red ? 1 : 0)), dim3(BUS_BLOCK), 0, M, (const Ctrl*)c, red, (const double*)dv.rec, dv.EndCount, dv.SlotCode, dv.busc, dv.B, dv, h->baseMVA);
        if (overlap) HIPCHK(hipEventRecord(e_bus, M));
        if (j + 1 < j0 + n) launch_int(j + 1);
        if (overlap) HIPCHK(hipStreamWaitEvent(X, e_bus, 0));
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool sample = d->xev_armed && j == j0 + std::min(1, n - 1) && hipStreamIsCapturing(X, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
        if (sample) {   // (once per arming, never inside a capture)
            d->xev_armed = false;
            d->xev_recorded = true;
            (void)hipEventRecord(d->xev[0], X);
        }
        if ((rc = dist_exchange(h, dv, true, c, nullptr, has_sums ? 0 : 2, X))) break;
        if (sample) (void)hipEventRecord(d->xev[1], X);
        if (j + 1 < j0 + n) launch_cut(j + 1);
    }
    // join (also after a failed exchange: a capture in progress must see X return to M)
    if (overlap) {
        (void)hipEventRecord(d->ev[1], X);
        (void)hipStreamWaitEvent(M, d->ev[1], 0);
    }
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

// Can the bound RCCL be captured?  Collective over the ranks, at connect time: one sums exchange enqueued directly (it also
// opens every peer connection before anything is captured), then the same exchange captured into a hipGraph, instantiated
// and replayed twice.  A rank whose capture fails still takes part in the two exchanges (enqueued directly: the peers' sends
// and receives match either way) and keeps enqueuing its batches directly from then on.  Every exchange is checked:
// rank r contributes (1, 2, 3, 4) (r + 1).
static int dist_probe(exaadmm_handle* h)
{
    exaadmm_dist* d = h->dist;
    d->graph_ok = false;
    d->probe_result = -1;   // not run (one rank / no communicator)
    if (d->nranks <= 1 || d->comm == nullptr) return EXAADMM_OK;
    d->probe_result = 0;
    hipStream_t X = d->xstream;
    const double mine[4] = {1.0 * (d->rank + 1), 2.0 * (d->rank + 1), 3.0 * (d->rank + 1), 4.0 * (d->rank + 1)};
    const double tri = 0.5 * d->nranks * (d->nranks + 1);
    double *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void**)&d_in, 4 * sizeof(double)));
    HIPCHK(hipMalloc((void**)&d_out, 4 * sizeof(double)));
    struct Free {
        double *a, *b;
        ~Free() { (void)hipFree(a); (void)hipFree(b); }
    } free_on_exit{d_in, d_out};
    HIPCHK(hipMemcpy(d_in, mine, sizeof(mine), hipMemcpyHostToDevice));
    auto one = [&]() -> int {
        hipLaunchKernelGGL(k_halo_pack_sums, dim3(1), dim3(64), 0, X, h->d, d_in, (const Ctrl*)nullptr);
        return dist_exchange(h, h->d, false, nullptr, d_out, 0, X);
    };
    auto verify = [&](const char* what) -> int {
        double got[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(got, d_out, sizeof(got), hipMemcpyDeviceToHost, X));
        HIPCHK(hipStreamSynchronize(X));
        for (int j = 0; j < 4; j++)
            if (got[j] != (j + 1) * tri) {
                set_err("exchange probe (%s): sum %d is %g, expected %g", what, j, got[j], (j + 1) * tri);
                return EXAADMM_ERR_HIP;
            }
        HIPCHK(hipMemsetAsync(d_out, 0, 4 * sizeof(double), X));
        return 0;
    };
    int rc;
    if ((rc = one())) return rc;
    if ((rc = verify("direct"))) return rc;
    hipGraphExec_t exec = nullptr;
    const bool want = h->use_graph;   // (option "graph" = 0: A/B switch, every batch enqueued directly)
    if (want && hipStreamBeginCapture(X, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int crc = one();
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(X, &g);
        if (crc == 0 && e == hipSuccess && g != nullptr && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
        if (crc != 0 || e != hipSuccess) exec = nullptr;
        if (g) (void)hipGraphDestroy(g);
    }
    (void)hipGetLastError();
    for (int rep = 0; rep < 2; rep++) {
        if (exec) {
            if (hipGraphLaunch(exec, X) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipGraphExecDestroy(exec);
                exec = nullptr;
                if ((rc = one())) return rc;
            } else
                g_exchanges++;
        } else if ((rc = one()))
            return rc;
        if ((rc = verify(exec ? "captured" : "direct, capture refused"))) {
            if (exec) (void)hipGraphExecDestroy(exec);
            return rc;
        }
    }
    d->graph_ok = (exec != nullptr);
    d->probe_result = d->graph_ok ? 1 : 0;
    if (exec) (void)hipGraphExecDestroy(exec);
    g_err.clear();   // (a library that refused the capture has left its message: handled here, not an error of the attach)
    return EXAADMM_OK;
}

// Exercises every RCCL entry point the loop uses, on the handle's stream, with whatever communicator is attached: the
// sums exchange (every rank contributes (1,2,3,4) * (rank+1): expected total = (1,2,3,4) * nranks (nranks+1) / 2) and a
// ring send / receive of n doubles (one rank: to itself).  *max_err = largest deviation from the expected values.
// Proves that the bound RCCL and this library share one HIP runtime (a stream handle of a foreign runtime would fail here).
extern "C" int exaadmm_dist_selftest(exaadmm_handle* h, int64_t n, double* max_err)
{
    REQUIRE(h && h->dist && h->dist->comm && n >= 1 && max_err, "needs an attached communicator");
    exaadmm_dist* d = h->dist;
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> a((size_t)n), b((size_t)n, -1.0);
    for (int64_t i = 0; i < n; i++) a[(size_t)i] = 0.5 * (double)i + d->rank;
    double *da = nullptr, *db = nullptr;
    struct Free {
        double *&a, *&b;
        ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
    } free_on_exit{da, db};
    HIPCHK(hipMalloc((void**)&da, (size_t)n * sizeof(double)));
    HIPCHK(hipMalloc((void**)&db, (size_t)n * sizeof(double)));
    HIPCHK(hipMemcpyAsync(da, a.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    double e = 0.0;
    if (d->nranks > 1) {
        const double mine[4] = {1.0 * (d->rank + 1), 2.0 * (d->rank + 1), 3.0 * (d->rank + 1), 4.0 * (d->rank + 1)};
        HIPCHK(hipMemcpyAsync(h->d_sums, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_halo_pack_sums, dim3(1), dim3(64), 0, h->stream, h->d, h->d_sums, (const Ctrl*)nullptr);
        int rc = dist_exchange(h, h->d, false, nullptr, h->d_sums, 0, h->stream);
        if (rc) return rc;
        double got[4];
        HIPCHK(hipMemcpyAsync(got, h->d_sums, sizeof(got), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if ((rc = dist_check_exchange(h))) return rc;   // (through the mailboxes: a peer that never answered)
        const double tri = 0.5 * d->nranks * (d->nranks + 1);
        for (int j = 0; j < 4; j++) {
            const double dev = fabs(got[j] - (j + 1) * tri);
            if (!(dev <= e)) e = dev;   // (a NaN is a deviation: fmax would drop it)
        }
    }
    const int next = (d->rank + 1) % d->nranks, prev = (d->rank + d->nranks - 1) % d->nranks;
    NCCLCHK(d, d->api.GroupStart());
    const ncclResult_t r1 = d->api.Send(da, (size_t)n, ncclDouble, next, d->comm, h->stream);
    const ncclResult_t r2 = d->api.Recv(db, (size_t)n, ncclDouble, prev, d->comm, h->stream);
    const ncclResult_t r3 = d->api.GroupEnd();
    REQUIRE(r1 == ncclSuccess && r2 == ncclSuccess && r3 == ncclSuccess, "ring send / receive failed");
    HIPCHK(hipMemcpyAsync(b.data(), db, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < n; i++) {
        const double dev = fabs(b[(size_t)i] - (0.5 * (double)i + prev));
        if (!(dev <= e)) e = dev;
    }
    *max_err = e;
    return EXAADMM_OK;
}

// The ring send / receive of the self test CAPTURED into a hipGraph and replayed twice on the exchange stream (one rank: to
// itself): is the bound RCCL capturable at all in this runtime?  *captured = 1: captured, instantiated, replayed, data correct;
// 0: the library refused (the loop then enqueues directly).  An error is returned only for wrong data.
extern "C" int exaadmm_dist_capture_selftest(exaadmm_handle* h, int64_t n, int* captured)
{
    REQUIRE(h && h->dist && h->dist->comm && n >= 1 && captured, "needs an attached communicator");
    exaadmm_dist* d = h->dist;
    *captured = 0;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t X = d->xstream;
    if (!X) HIPCHK(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
    X = d->xstream;
    std::vector<double> a((size_t)n), b((size_t)n, -1.0);
    double *da = nullptr, *db = nullptr;
    struct Free {
        double *&a, *&b;
        ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
    } free_on_exit{da, db};
    HIPCHK(hipMalloc((void**)&da, (size_t)n * sizeof(double)));
    HIPCHK(hipMalloc((void**)&db, (size_t)n * sizeof(double)));
    const int next = (d->rank + 1) % d->nranks, prev = (d->rank + d->nranks - 1) % d->nranks;
    auto ring = [&]() -> bool {
        if (d->api.GroupStart() != ncclSuccess) return false;
        const ncclResult_t r1 = d->api.Send(da, (size_t)n, ncclDouble, next, d->comm, X);
        const ncclResult_t r2 = r1 == ncclSuccess ? d->api.Recv(db, (size_t)n, ncclDouble, prev, d->comm, X) : r1;
        const ncclResult_t r3 = d->api.GroupEnd();
        return r1 == ncclSuccess && r2 == ncclSuccess && r3 == ncclSuccess;
    };
    // (first use of the connection outside any capture)
    for (int64_t i = 0; i < n; i++) a[(size_t)i] = 0.25 * (double)i + d->rank;
    HIPCHK(hipMemcpy(da, a.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    REQUIRE(ring(), "ring send / receive failed");
    HIPCHK(hipStreamSynchronize(X));
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(X, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const bool ok = ring();
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(X, &g);
        if (ok && e == hipSuccess && g != nullptr && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
        if (!ok || e != hipSuccess) exec = nullptr;
        if (g) (void)hipGraphDestroy(g);
    }
    (void)hipGetLastError();
    for (int rep = 1; rep <= 2; rep++) {
        for (int64_t i = 0; i < n; i++) a[(size_t)i] = 0.25 * (double)i + d->rank + 100.0 * rep;
        HIPCHK(hipMemcpyAsync(da, a.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, X));
        HIPCHK(hipMemsetAsync(db, 0, (size_t)n * sizeof(double), X));
        if (exec) {
            if (hipGraphLaunch(exec, X) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipGraphExecDestroy(exec);
                exec = nullptr;
            }
        }
        if (!exec) REQUIRE(ring(), "ring send / receive failed");
        HIPCHK(hipMemcpyAsync(b.data(), db, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, X));
        HIPCHK(hipStreamSynchronize(X));
        for (int64_t i = 0; i < n; i++)
            if (b[(size_t)i] != 0.25 * (double)i + prev + 100.0 * rep) {
                if (exec) (void)hipGraphExecDestroy(exec);
                set_err("captured ring send / receive delivered wrong data at %lld (replay %d)", (long long)i, rep);
                return EXAADMM_ERR_HIP;
            }
    }
    *captured = exec ? 1 : 0;
    if (exec) (void)hipGraphExecDestroy(exec);
    return EXAADMM_OK;
}
