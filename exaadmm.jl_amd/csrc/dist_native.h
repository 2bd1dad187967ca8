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
};

static RcclApi g_rccl;   // process-wide function table
static std::string g_rccl_path;
static std::mutex g_rccl_mutex;   // handles may be attached from different host threads

// which RCCL the library bound (empty before the first exaadmm_dist_unique_id / multi-rank exaadmm_dist_prepare)
extern "C" const char* exaadmm_dist_rccl_path(void) { return g_rccl_path.c_str(); }

static int rccl_bind(RcclApi* out)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.lib) {
        std::vector<std::string> names;
        if (const char* e = getenv("EXAADMM_RCCL_LIB")) names.push_back(e);
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
            set_err("RCCL not found next to the HIP runtime (set EXAADMM_RCCL_LIB): %s", dlerror());
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

static bool dist_has_peers(const exaadmm_handle* h) { return h->dist != nullptr && h->dist->nranks > 1; }
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
    if (d->comm) (void)d->api.CommDestroy(d->comm);
    void* bufs[] = {d->send, d->recv, d->local, d->d_pos, d->d_off, d->d_rcode, d->d_rpos};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
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

// number of grouped exchanges this handle has enqueued since it was attached (tests: exactly one per inner iteration)
static thread_local int64_t g_exchanges = 0;
extern "C" int64_t exaadmm_dist_exchange_count(void) { return g_exchanges; }

// The one exchange: per peer one send and one receive of (its cut branches' v entries +) the four sums, grouped; then
// the unpack kernel.  with_v = false: the sums only (end of an inner loop, ||z|| of the outer loop).
static int dist_exchange(exaadmm_handle* h, const Dev& dv, bool with_v, Ctrl* c, double* out, int need_pending)
{
    exaadmm_dist* d = h->dist;
    REQUIRE(d->comm != nullptr, "partition has peers but no communicator (exaadmm_dist_connect)");
    NCCLCHK(d, d->api.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int p = 0; p < d->nranks && bad == ncclSuccess; p++) {
        if (p == d->rank) continue;
        const size_t off = (size_t)(with_v ? d->msg_off[p] : d->msg_off[p] + 4 * d->ncut[p]);
        const size_t n = (size_t)(with_v ? 4 * d->ncut[p] + 4 : 4);
        bad = d->api.Send(d->send + off, n, ncclDouble, p, d->comm, h->stream);
        if (bad == ncclSuccess) bad = d->api.Recv(d->recv + off, n, ncclDouble, p, d->comm, h->stream);
    }
    const ncclResult_t ge = d->api.GroupEnd();   // always closed, also after a failed call inside the group
    if (bad != ncclSuccess || ge != ncclSuccess) {
        set_err("halo exchange failed: %s", d->api.GetErrorString(bad != ncclSuccess ? bad : ge));
        return EXAADMM_ERR_HIP;
    }
    g_exchanges++;
    const int nc = with_v ? d->n_cut : 0;
    hipLaunchKernelGGL(k_halo_unpack, grid_for(std::max(nc, 1), 256), dim3(256), 0, h->stream, dv, d->d_rcode, d->d_rpos, nc, c, out, need_pending);
    return 0;
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
        int rc = dist_exchange(h, h->d, false, nullptr, h->d_sums, 0);
        if (rc) return rc;
        double got[4];
        HIPCHK(hipMemcpyAsync(got, h->d_sums, sizeof(got), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const double tri = 0.5 * d->nranks * (d->nranks + 1);
        for (int j = 0; j < 4; j++) e = fmax(e, fabs(got[j] - (j + 1) * tri));
    }
    const int next = (d->rank + 1) % d->nranks, prev = (d->rank + d->nranks - 1) % d->nranks;
    NCCLCHK(d, d->api.GroupStart());
    const ncclResult_t r1 = d->api.Send(da, (size_t)n, ncclDouble, next, d->comm, h->stream);
    const ncclResult_t r2 = d->api.Recv(db, (size_t)n, ncclDouble, prev, d->comm, h->stream);
    const ncclResult_t r3 = d->api.GroupEnd();
    REQUIRE(r1 == ncclSuccess && r2 == ncclSuccess && r3 == ncclSuccess, "ring send / receive failed");
    HIPCHK(hipMemcpyAsync(b.data(), db, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < n; i++) e = fmax(e, fabs(b[(size_t)i] - (0.5 * (double)i + prev)));
    *max_err = e;
    return EXAADMM_OK;
}
