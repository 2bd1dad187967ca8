// TEST INFRASTRUCTURE ONLY -- an in-process stand-in for the eight RCCL entry points csrc/dist_native.h binds, so that
// the library's distributed inner loop can be executed with SEVERAL ranks on a ONE-GPU box: every rank is a host
// thread of the same process driving its own handle (its own stream) on the same device; "ranks" rendezvous through
// mutex-protected mailboxes and move data with device-to-device copies.  (Real RCCL refuses two ranks on one device.)
// Semantics kept: ncclSend / ncclRecv inside a group are matched per (source, destination) pair in posting order, a
// rank need not take part in a group it has no messages in.  Everything is blocking on the host -- slower than RCCL,
// same results.  rccl_shim_counts() reports how many groups / sends / receives were issued.
// Not capturable: a send / receive on a stream that is being captured is refused (the library must then enqueue its batches
// directly).  RCCL_SHIM_DELAY_US=n: every message becomes visible to its receiver n microseconds after it was posted -- a
// slow interconnect; the library's overlap of the exchange with the interior branch solves is measured against it.
// RCCL_SHIM_ASYNC=1: stream-ordered mode.  Nothing blocks on the host: a send is a small kernel on the caller's stream that
// copies the message into a per-(source, destination) mailbox in device memory and raises its sequence flag, a receive is a
// kernel that waits for the flag (bounded spin, one lane), copies the message out and acknowledges.  Sequence numbers live in
// device memory and are advanced by the kernels themselves, so the calls can be CAPTURED into a hipGraph and replayed -- like
// RCCL's, and unlike the host-blocking mode, whose ~80 us of host synchronisation per exchange would hide what the library's
// overlap of the exchange with the interior branch solves does.  The delay option applies on the device (the receiver waits
// until the message is n microseconds old).  A wait that runs into its cap sets a sticky failure flag (rccl_shim_failures).
// Only dependable with TWO ranks: a receive kernel that spins at the head of a hardware queue blocks whatever the runtime has
// put behind it in that queue, and with three thread-ranks (six streams + the graphs' internal ones on one device) a peer's
// send does end up behind somebody's receive -- observed as a multi-second stall.  Real RCCL has a queue per process.
// Selected with EXAADMM_RCCL_LIB=<this .so>; built and used by tests/test_gpu_dist_threads.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Msg {
    const void* src;
    size_t bytes;
    bool consumed = false;
    std::chrono::steady_clock::time_point visible;
};
long long delay_us()
{
    static const long long v = getenv("RCCL_SHIM_DELAY_US") ? atoll(getenv("RCCL_SHIM_DELAY_US")) : 0;
    return v;
}
bool capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}
bool async_mode()
{
    static const bool v = getenv("RCCL_SHIM_ASYNC") != nullptr && atoi(getenv("RCCL_SHIM_ASYNC")) != 0;
    return v;
}
// stream-ordered mode: one mailbox per ordered pair of ranks
constexpr size_t BOX_DOUBLES = 1 << 17;   // 1 MiB per mailbox
struct Box {
    double* data;
    // [0] sequence number posted, [1] sequence number consumed, [2] wall clock of the post, [3] sends issued, [4] receives issued
    unsigned long long* hdr;
};
struct World {
    int nranks = 0, attached = 0;
    std::map<std::pair<int, int>, std::deque<Msg*>> box;   // (from, to) -> posted sends
    std::vector<Box> abox;                                  // async mode: [from * nranks + to]
    unsigned long long* fail = nullptr;
};
unsigned long long* g_fail = nullptr;

__global__ void __launch_bounds__(256) k_shim_send(Box b, const double* src, size_t n, unsigned long long* fail)
{
    __shared__ unsigned long long seq;
    if (threadIdx.x == 0) {
        seq = b.hdr[3] + 1;          // (only this mailbox's sender touches word 3, in stream order)
        b.hdr[3] = seq;
        int spins = 0;               // the previous message must have been consumed before the mailbox is overwritten
        while (__hip_atomic_load(&b.hdr[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) + 1 < seq) {
            // (the peer's host thread may be busy for a long time -- Python ranks share one interpreter lock, a rank may be
            //  instantiating a graph: the cap is seconds, not milliseconds)
            if (++spins > (1 << 23) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                unsigned long long expect = 0ull;
                __hip_atomic_compare_exchange_strong(fail, &expect, (1ull << 56) | seq, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) b.data[i] = src[i];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        b.hdr[2] = wall_clock64();
        __hip_atomic_store(&b.hdr[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256) k_shim_recv(Box b, double* dst, size_t n, unsigned long long delay_ticks, unsigned long long* fail)
{
    __shared__ unsigned long long seq;
    if (threadIdx.x == 0) {
        seq = b.hdr[4] + 1;
        b.hdr[4] = seq;
        int spins = 0;
        while (__hip_atomic_load(&b.hdr[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
            if (++spins > (1 << 23) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                unsigned long long expect = 0ull;
                __hip_atomic_compare_exchange_strong(fail, &expect, (2ull << 56) | seq, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        if (delay_ticks) {   // a slow interconnect: the message becomes visible delay_ticks after it was posted
            const unsigned long long t_post = __hip_atomic_load(&b.hdr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (wall_clock64() - t_post < delay_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = b.data[i];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&b.hdr[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
struct Comm {
    World* w;
    int rank;
    bool in_group = false;
    struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };
    std::vector<Op> ops;
};
std::mutex g_mu;
std::condition_variable g_cv;
long long g_counts[3] = {0, 0, 0};
std::map<unsigned long long, World*> g_worlds;
unsigned long long g_next_id = 1;

ncclResult_t run_ops_async(Comm* c)
{
    World* w = c->w;
    const unsigned long long delay_ticks = (unsigned long long)delay_us() * 100ull;   // wall_clock64: 100 MHz
    for (int pass = 0; pass < 2; pass++)   // all sends first: a send never waits for anything of the current round
        for (auto& op : c->ops) {
            if (op.send != (pass == 0)) continue;
            if (op.bytes / sizeof(double) > BOX_DOUBLES) return ncclInvalidArgument;
            if (op.send)
                hipLaunchKernelGGL(k_shim_send, dim3(1), dim3(256), 0, op.stream, w->abox[(size_t)c->rank * w->nranks + op.peer], (const double*)op.buf,
                                   op.bytes / sizeof(double), w->fail);
            else
                hipLaunchKernelGGL(k_shim_recv, dim3(1), dim3(256), 0, op.stream, w->abox[(size_t)op.peer * w->nranks + c->rank], (double*)op.buf,
                                   op.bytes / sizeof(double), delay_ticks, w->fail);
        }
    c->ops.clear();
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t run_ops(Comm* c)
{
    if (async_mode()) return run_ops_async(c);
    World* w = c->w;
    std::vector<Msg*> mine;
    for (auto& op : c->ops)
        if (op.send) {
            if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;   // the data is final
            Msg* m = new Msg{op.buf, op.bytes};
            m->visible = std::chrono::steady_clock::now() + std::chrono::microseconds(delay_us());
            mine.push_back(m);
            std::lock_guard<std::mutex> lk(g_mu);
            w->box[{c->rank, op.peer}].push_back(m);
            g_cv.notify_all();
        }
    for (auto& op : c->ops)
        if (!op.send) {
            Msg* m = nullptr;
            {
                std::unique_lock<std::mutex> lk(g_mu);
                auto& q = w->box[{op.peer, c->rank}];
                g_cv.wait(lk, [&] { return !q.empty(); });
                m = q.front();
                q.pop_front();
            }
            if (m->bytes != op.bytes) return ncclInvalidArgument;
            while (std::chrono::steady_clock::now() < m->visible) {}   // (busy wait: microseconds)
            // (a device-to-device hipMemcpy may return before the copy has run: copy on the receiving stream and wait
            //  for it, only then may the sender reuse its buffer)
            if (hipMemcpyAsync(op.buf, m->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
            std::lock_guard<std::mutex> lk(g_mu);
            m->consumed = true;
            g_cv.notify_all();
        }
    for (Msg* m : mine) {   // the send buffer may be reused only after the receiver has copied it
        std::unique_lock<std::mutex> lk(g_mu);
        g_cv.wait(lk, [&] { return m->consumed; });
        lk.unlock();
        delete m;
    }
    c->ops.clear();
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof(*id));
    const unsigned long long k = g_next_id++;
    memcpy(id->internal, &k, sizeof(k));
    g_worlds[k] = new World();
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    unsigned long long k;
    memcpy(&k, id.internal, sizeof(k));
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = g_worlds.find(k);
    if (it == g_worlds.end()) return ncclInvalidArgument;
    World* w = it->second;
    if (w->nranks == 0) {
        w->nranks = nranks;
        if (async_mode()) {   // the first rank to arrive builds the mailboxes
            if (hipMalloc((void**)&w->fail, 8) != hipSuccess || hipMemset(w->fail, 0, 8) != hipSuccess) return ncclUnhandledCudaError;
            g_fail = w->fail;
            w->abox.resize((size_t)nranks * nranks);
            for (auto& b : w->abox) {
                if (hipMalloc((void**)&b.data, BOX_DOUBLES * sizeof(double)) != hipSuccess || hipMalloc((void**)&b.hdr, 64) != hipSuccess ||
                    hipMemset(b.hdr, 0, 64) != hipSuccess)
                    return ncclUnhandledCudaError;
            }
            if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    if (w->nranks != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    w->attached++;
    g_cv.notify_all();
    g_cv.wait(lk, [&] { return w->attached >= nranks; });   // collective, like the real call
    Comm* c = new Comm{w, rank};
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { return ncclSuccess; }   // ops carry their communicator: grouping is per communicator below
ncclResult_t ncclGroupEnd();
static thread_local Comm* t_group_comm = nullptr;
}
extern "C" ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    if (t != ncclDouble) return ncclInvalidArgument;
    if (!async_mode() && capturing(s)) return ncclInvalidUsage;
    Comm* c = reinterpret_cast<Comm*>(comm);
    c->ops.push_back({true, const_cast<void*>(buf), count * sizeof(double), peer, s});
    t_group_comm = c;
    return ncclSuccess;
}
extern "C" ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    if (t != ncclDouble) return ncclInvalidArgument;
    if (!async_mode() && capturing(s)) return ncclInvalidUsage;
    Comm* c = reinterpret_cast<Comm*>(comm);
    c->ops.push_back({false, buf, count * sizeof(double), peer, s});
    t_group_comm = c;
    return ncclSuccess;
}
extern "C" ncclResult_t ncclGroupEnd()
{
    Comm* c = t_group_comm;
    t_group_comm = nullptr;
    if (!c) return ncclSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_counts[0] += 1;
        for (auto& op : c->ops) g_counts[op.send ? 1 : 2] += 1;
    }
    return run_ops(c);
}
// call counters for the tests: {groups that carried operations, sends, receives}.  (There is no ncclAllReduce here on
// purpose: the library's loop must not need one.)
extern "C" void rccl_shim_counts(long long out[3])
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < 3; i++) out[i] = g_counts[i];
}
// async mode: has any device-side wait run into its cap?  0 = no; else (1 = send waiting for the previous message's
// acknowledgement, 2 = receive waiting for the message) << 56 | sequence number of the first one that did.  (Synchronises the device.)
extern "C" long long rccl_shim_failures(void)
{
    if (!g_fail) return 0;
    unsigned long long v = 0;
    if (hipMemcpy(&v, g_fail, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)v;
}
// async mode, diagnostics: per mailbox (from, to) the five header words {posted, consumed, post time, sends issued, receives issued}
extern "C" int rccl_shim_dump(unsigned long long* out, int max_boxes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto& kv : g_worlds)
        for (auto& b : kv.second->abox) {
            if (n >= max_boxes) return n;
            if (hipMemcpy(out + 5 * (size_t)n, b.hdr, 40, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            n++;
        }
    return n;
}
extern "C" const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "rccl_shim error"; }
