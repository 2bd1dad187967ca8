// Distributed inner loop inside the library: one handle = one partition on one MI355X, one process per GPU,
// halo exchanges and the residual reduction issued to RCCL from C++ on the handle's own stream.
// (Included at the end of exaadmm_hip.hip: uses the handle and the launch helpers defined there.)
//
// The reference has no multi-process path (acopf_model.jl:49-56 is vestigial scaffolding); the scheme is
// per inner iteration
//     x-update (generators + native branches)
//     halo A : u, z, l of the 4 to-side entries of every cut branch -> owner of the to-bus     (grouped send/recv)
//     bus update
//     halo B : the 4 new v entries back to the branch owner                                     (grouped send/recv)
//     fused z / l / residual on owned elements -> 4 partial sums of squares
//     all-reduce of the 4 sums (32 bytes), loop control on the device (k_ctrl_step)
// Round 1 drove this sequence from Python (ten ctypes calls and three torch collectives per iteration: host-bound,
// 0.55x of the fused loop at one rank).  Here an iteration is nine kernel launches and at most three RCCL calls from
// C++, enqueued in batches of eight with ONE host round trip per batch, as exaadmm_run_inner does on one GPU.
//
// RCCL is bound at run time (dlopen) so that the library keeps loading on hosts without it.  It must be the RCCL that
// sits on THE SAME HIP runtime as this library, because the handle's stream is handed to it: a Python process that
// also imports PyTorch holds a second, private HIP runtime and RCCL (torch/lib/), whose stream handles mean nothing to
// this one.  The copy next to the libamdhip64 this library is bound to (normally /opt/rocm/lib) is therefore loaded by
// path; torch.distributed is only used by the caller to hand the 128-byte id around.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>

struct exaadmm_dist {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    int64_t nvar_global = 0;
    std::vector<int64_t> cnt[4];   // per-peer element counts: send u/z/l, recv u/z/l, send v, recv v
    double* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    bool any_peer = false;
};

static exaadmm_dist g_rccl_binding;   // function table only (comm stays per handle)
static std::string g_rccl_path;

// which RCCL the library bound (empty before the first exaadmm_dist_unique_id / multi-rank exaadmm_dist_attach)
extern "C" const char* exaadmm_dist_rccl_path(void) { return g_rccl_path.c_str(); }

static std::mutex g_rccl_mutex;   // handles may be attached from different host threads

static int rccl_bind(exaadmm_dist* d)
{
    if (d->lib) return 0;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl_binding.lib) {
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
        for (const std::string& n : names)
            if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) {
                g_rccl_path = n;
                break;
            }
        if (!lib) {
            set_err("RCCL not found next to the HIP runtime (set EXAADMM_RCCL_LIB): %s", dlerror());
            return EXAADMM_ERR_INVALID;
        }
        exaadmm_dist& b = g_rccl_binding;
        b.lib = lib;
#define EXA_SYM(field, name)                                         \
    do {                                                             \
        *(void**)(&b.field) = dlsym(lib, name);                      \
        if (!b.field) {                                              \
            set_err("RCCL symbol %s missing", name);                 \
            b.lib = nullptr;                                         \
            return EXAADMM_ERR_INVALID;                              \
        }                                                            \
    } while (0)
        EXA_SYM(GetUniqueId, "ncclGetUniqueId");
        EXA_SYM(CommInitRank, "ncclCommInitRank");
        EXA_SYM(CommDestroy, "ncclCommDestroy");
        EXA_SYM(GroupStart, "ncclGroupStart");
        EXA_SYM(GroupEnd, "ncclGroupEnd");
        EXA_SYM(Send, "ncclSend");
        EXA_SYM(Recv, "ncclRecv");
        EXA_SYM(AllReduce, "ncclAllReduce");
        EXA_SYM(GetErrorString, "ncclGetErrorString");
#undef EXA_SYM
    }
    const exaadmm_dist& b = g_rccl_binding;
    d->lib = b.lib;
    d->GetUniqueId = b.GetUniqueId; d->CommInitRank = b.CommInitRank; d->CommDestroy = b.CommDestroy;
    d->GroupStart = b.GroupStart; d->GroupEnd = b.GroupEnd; d->Send = b.Send; d->Recv = b.Recv;
    d->AllReduce = b.AllReduce; d->GetErrorString = b.GetErrorString;
    return 0;
}

#define NCCLCHK(d, expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess) {                                                                          \
            set_err("%s failed: %s (%s:%d)", #expr, (d)->GetErrorString(r__), __FILE__, __LINE__);         \
            return EXAADMM_ERR_HIP;                                                                        \
        }                                                                                                  \
    } while (0)

extern "C" int exaadmm_dist_unique_id(void* out, int64_t nbytes)
{
    REQUIRE(out && nbytes == (int64_t)sizeof(ncclUniqueId), "buffer must hold 128 bytes (ncclUniqueId)");
    exaadmm_dist tmp;
    int rc = rccl_bind(&tmp);
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(&tmp, tmp.GetUniqueId(&id));
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
    if (d->comm) (void)d->CommDestroy(d->comm);
    for (double* b : d->buf)
        if (b) (void)hipFree(b);
    delete d;
    h->dist = nullptr;
    return EXAADMM_OK;
}

// counts: 4 x nranks, row-major: [send u/z/l per peer | recv u/z/l per peer | send v per peer | recv v per peer],
// in doubles, the per-peer block order of the halo index lists in slots 0..3 (exaadmm_set_halo_index).
// nranks == 1 needs no communicator: id may be NULL.
static int dist_attach_impl(exaadmm_handle* h, int nranks, int rank, const void* unique_id, int64_t id_bytes, int64_t nvar_global,
                            const int64_t* counts);

extern "C" int exaadmm_dist_attach(exaadmm_handle* h, int nranks, int rank, const void* unique_id, int64_t id_bytes,
                                   int64_t nvar_global, const int64_t* counts)
{
    const int rc = dist_attach_impl(h, nranks, rank, unique_id, id_bytes, nvar_global, counts);
    if (rc != EXAADMM_OK && h != nullptr && h->dist != nullptr) {   // no half-attached handle survives a failure
        const std::string keep = g_err;
        (void)exaadmm_dist_detach(h);
        g_err = keep;
    }
    return rc;
}

static int dist_attach_impl(exaadmm_handle* h, int nranks, int rank, const void* unique_id, int64_t id_bytes, int64_t nvar_global,
                            const int64_t* counts)
{
    REQUIRE(h && nranks >= 1 && rank >= 0 && rank < nranks && nvar_global >= 1 && counts, "bad argument");
    REQUIRE(h->T == 1, "partition helpers are not available on a multi-period handle");
    HIPCHK(hipSetDevice(h->device));
    int rc = exaadmm_dist_detach(h);
    if (rc) return rc;
    exaadmm_dist* d = new exaadmm_dist();
    h->dist = d;
    d->nranks = nranks;
    d->rank = rank;
    d->nvar_global = nvar_global;
    for (int s = 0; s < 4; s++) {
        d->cnt[s].assign(counts + (size_t)s * nranks, counts + (size_t)(s + 1) * nranks);
        int64_t tot = 0;
        for (int p = 0; p < nranks; p++) {
            REQUIRE(d->cnt[s][p] >= 0, "negative halo count");
            REQUIRE(p != rank || d->cnt[s][p] == 0, "a rank has no halo with itself");
            tot += d->cnt[s][p];
            if (d->cnt[s][p] > 0) d->any_peer = true;
        }
        REQUIRE(tot == h->halo[s].count, "halo counts do not add up to the index list of the slot (exaadmm_set_halo_index)");
        if (tot > 0) HIPCHK(hipMalloc((void**)&d->buf[s], (size_t)tot * sizeof(double)));
    }
    if (nranks > 1 || unique_id != nullptr) {   // (a one-rank communicator is legal: used by the self test)
        REQUIRE(unique_id && id_bytes == (int64_t)sizeof(ncclUniqueId), "unique id must be the 128 bytes of exaadmm_dist_unique_id");
        if ((rc = rccl_bind(d))) return rc;
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        NCCLCHK(d, d->CommInitRank(&d->comm, nranks, id, rank));
    }
    return EXAADMM_OK;
}

static int dist_exchange(exaadmm_handle* h, int send_slot, int recv_slot)
{
    exaadmm_dist* d = h->dist;
    const HaloSlot& ss = h->halo[send_slot];
    const HaloSlot& rs = h->halo[recv_slot];
    if (ss.count > 0)
        hipLaunchKernelGGL(k_gather, grid_for(ss.count, 256), dim3(256), 0, h->stream, vec_table(h), ss.d_which, ss.d_idx, ss.count, d->buf[send_slot]);
    if (d->nranks > 1 && (ss.count > 0 || rs.count > 0)) {
        REQUIRE(d->comm != nullptr, "partition has peers but no communicator (exaadmm_dist_attach failed?)");
        NCCLCHK(d, d->GroupStart());
        int64_t so = 0, ro = 0;
        for (int p = 0; p < d->nranks; p++) {
            const int64_t ns = d->cnt[send_slot][p], nr = d->cnt[recv_slot][p];
            if (ns > 0) NCCLCHK(d, d->Send(d->buf[send_slot] + so, (size_t)ns, ncclDouble, p, d->comm, h->stream));
            if (nr > 0) NCCLCHK(d, d->Recv(d->buf[recv_slot] + ro, (size_t)nr, ncclDouble, p, d->comm, h->stream));
            so += ns;
            ro += nr;
        }
        NCCLCHK(d, d->GroupEnd());
    }
    if (rs.count > 0)
        hipLaunchKernelGGL(k_scatter, grid_for(rs.count, 256), dim3(256), 0, h->stream, vec_table(h), rs.d_which, rs.d_idx, rs.count, d->buf[recv_slot]);
    return 0;
}

static int dist_allreduce_sums(exaadmm_handle* h)
{
    exaadmm_dist* d = h->dist;
    if (d->comm) NCCLCHK(d, d->AllReduce(h->d_sums, h->d_sums, 4, ncclDouble, ncclSum, d->comm, h->stream));
    return 0;
}

// Exercises every RCCL entry point the loop uses, on the handle's stream, with whatever communicator is attached
// (one rank is enough: a rank may send to itself inside a group): all-reduce of 4 doubles, grouped send + receive of
// n doubles.  *max_err = largest deviation from the expected values.  Proves that the bound RCCL and this library
// share one HIP runtime (a stream handle of a foreign runtime would fail here).
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
    const double four[4] = {1.0, 2.0, 3.0, 4.0};
    HIPCHK(hipMemcpyAsync(h->d_sums, four, sizeof(four), hipMemcpyHostToDevice, h->stream));
    NCCLCHK(d, d->AllReduce(h->d_sums, h->d_sums, 4, ncclDouble, ncclSum, d->comm, h->stream));
    const int next = (d->rank + 1) % d->nranks, prev = (d->rank + d->nranks - 1) % d->nranks;
    NCCLCHK(d, d->GroupStart());
    NCCLCHK(d, d->Send(da, (size_t)n, ncclDouble, next, d->comm, h->stream));
    NCCLCHK(d, d->Recv(db, (size_t)n, ncclDouble, prev, d->comm, h->stream));
    NCCLCHK(d, d->GroupEnd());
    double got[4];
    HIPCHK(hipMemcpyAsync(got, h->d_sums, sizeof(got), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(b.data(), db, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double e = 0.0;
    for (int j = 0; j < 4; j++) e = fmax(e, fabs(got[j] - d->nranks * four[j]));
    for (int64_t i = 0; i < n; i++) e = fmax(e, fabs(b[(size_t)i] - (0.5 * (double)i + prev)));
    *max_err = e;
    return EXAADMM_OK;
}

// one inner iteration, enqueue only.  c != nullptr: armed device loop control (major_iter and beta come from it)
static int dist_enqueue_iteration(exaadmm_handle* h, Ctrl* c, int major_iter, double beta)
{
    int rc;
    launch_branch(h, major_iter, c, true);
    if ((rc = dist_exchange(h, 0, 1))) return rc;
    launch_bus(h, c);
    if ((rc = dist_exchange(h, 2, 3))) return rc;
    launch_fused_zl(h, beta, c);
    if (h->dist->comm == nullptr) {
        launch_final(h, c, h->d_sums);   // one partition, nothing to all-reduce: the reduction advances the loop control itself
        return 0;
    }
    launch_final(h, nullptr, h->d_sums);
    if ((rc = dist_allreduce_sums(h))) return rc;
    if (c) hipLaunchKernelGGL(k_ctrl_step, dim3(1), dim3(64), 0, h->stream, h->d_sums, c);
    return 0;
}

extern "C" int exaadmm_dist_run_inner(exaadmm_handle* h, int outer, int max_inner, double beta, int* inner_done, double out[4])
{
    REQUIRE(h && h->dist && outer >= 1 && max_inner >= 1, "bad argument (exaadmm_dist_attach first)");
    HIPCHK(hipSetDevice(h->device));
    Ctrl* hc = h->h_ctrl;
    memset(hc, 0, sizeof(Ctrl));
    hc->max_inner = max_inner;
    hc->check_every = 1;   // every rank must take the same decision in the same iteration
    hc->eps_pri = sqrt((double)h->dist->nvar_global) / (2500.0 * outer);   // admm_two_level.jl:45 on the GLOBAL nvar
    hc->beta = beta;
    HIPCHK(hipMemcpyAsync(h->d_ctrl, hc, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->residual_vectors_valid = false;
    if (h->dev_dirty) {
        HIPCHK(hipMemcpyAsync(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice, h->stream));
        h->dev_dirty = false;
    }
    const int batch = 8;
    int enq = 0, rc;
    while (enq < max_inner) {
        const int nb = std::min(batch, max_inner - enq);
        for (int i = 0; i < nb; i++)
            if ((rc = dist_enqueue_iteration(h, h->d_ctrl, 0, beta))) return rc;
        enq += nb;
        HIPCHK(hipMemcpyAsync(hc, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
        if ((rc = sync_check(h, true))) return rc;
        if (hc->done) break;   // the same all-reduced sums on every rank -> the same decision everywhere
    }
    if (inner_done) *inner_done = hc->inner;
    if (out)
        for (int j = 0; j < 4; j++) out[j] = hc->res[j];
    return EXAADMM_OK;
}

extern "C" int exaadmm_dist_run_fixed(exaadmm_handle* h, int iters, int first_major_iter, double beta, double out[4], double* ms_total)
{
    REQUIRE(h && h->dist && iters >= 1 && first_major_iter >= 1, "bad argument (exaadmm_dist_attach first)");
    HIPCHK(hipSetDevice(h->device));
    while (h->ev.size() < 2) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    h->residual_vectors_valid = false;
    if (h->dev_dirty) {
        HIPCHK(hipMemcpyAsync(h->d_dev, &h->d, sizeof(Dev), hipMemcpyHostToDevice, h->stream));
        h->dev_dirty = false;
    }
    int rc;
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    for (int i = 0; i < iters; i++)
        if ((rc = dist_enqueue_iteration(h, nullptr, first_major_iter + i, beta))) return rc;
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    double s[4];
    if ((rc = fetch_sums(h, s, true))) return rc;
    if (out)
        for (int j = 0; j < 4; j++) out[j] = s[j];
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    if (ms_total) *ms_total = ms;
    return EXAADMM_OK;
}

// admm_outer_prestep over all partitions: ||z_curr|| of the owned elements, all-reduced
extern "C" int exaadmm_dist_norm_z(exaadmm_handle* h, double* out)
{
    REQUIRE(h && h->dist && out, "bad argument (exaadmm_dist_attach first)");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_sumsq_z, dim3(h->ew_blocks), dim3(EW_BLOCK), 0, h->stream, h->d);
    launch_final(h, nullptr, h->d_sums);
    int rc = dist_allreduce_sums(h);
    if (rc) return rc;
    double s[4];
    if ((rc = fetch_sums(h, s, true))) return rc;
    *out = s[2];
    return EXAADMM_OK;
}
