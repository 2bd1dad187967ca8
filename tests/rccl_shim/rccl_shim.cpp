// TEST INFRASTRUCTURE ONLY -- an in-process stand-in for the eight RCCL entry points csrc/dist_native.h binds, so that
// the library's distributed inner loop can be executed with SEVERAL ranks on a ONE-GPU box: every rank is a host
// thread of the same process driving its own handle (its own stream) on the same device; "ranks" rendezvous through
// mutex-protected mailboxes and move data with device-to-device copies.  (Real RCCL refuses two ranks on one device.)
// Semantics kept: ncclSend / ncclRecv inside a group are matched per (source, destination) pair in posting order, a
// rank need not take part in a group it has no messages in.  Everything is blocking on the host -- slower than RCCL,
// same results.  rccl_shim_counts() reports how many groups / sends / receives were issued.
// Selected with EXAADMM_RCCL_LIB=<this .so>; built and used by tests/test_gpu_dist_threads.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
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
};
struct World {
    int nranks = 0, attached = 0;
    std::map<std::pair<int, int>, std::deque<Msg*>> box;   // (from, to) -> posted sends
};
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

ncclResult_t run_ops(Comm* c)
{
    World* w = c->w;
    std::vector<Msg*> mine;
    for (auto& op : c->ops)
        if (op.send) {
            if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;   // the data is final
            Msg* m = new Msg{op.buf, op.bytes};
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
    if (w->nranks == 0) w->nranks = nranks;
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
    Comm* c = reinterpret_cast<Comm*>(comm);
    c->ops.push_back({true, const_cast<void*>(buf), count * sizeof(double), peer, s});
    t_group_comm = c;
    return ncclSuccess;
}
extern "C" ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    if (t != ncclDouble) return ncclInvalidArgument;
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
extern "C" const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "rccl_shim error"; }
