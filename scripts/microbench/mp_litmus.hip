// Message-passing litmus test for the hand-overs of the resident loop (kernels.h: st_agent / ld_agent, resident_order_stores,
// resident_arrive, resident_wait) on MI355X: does the sequence the loop ships order DATA before FLAG for an observer on
// ANOTHER XCD, and do the sequences it was NOT allowed to use actually fail (positive controls)?
//
//   pairs mode: block 2k (writer) and block 2k+1 (reader) -- consecutive blocks land on different XCDs (round-robin over the
//   8 XCDs) -- play ping-pong for `iters` rounds; in round `it` the writer's 64 lanes publish 4 doubles each, f(it, lane, q),
//   then ONE lane publishes flag = it; the reader polls the flag, reads the 256 doubles, counts every value that is not
//   f(it, ...), and acknowledges.  The reader never reads data before it has seen the flag, so every mismatch is a
//   reordering (or a stale cache line) and nothing else.
//   arrive mode (the bus update's protocol): 4 blocks on 4 XCDs each publish their 4 terms of round `it` and count themselves in
//   with an agent-scope fetch_add; the block whose arrival completes the round (old + 1 == 4 it) reads all 16 terms, checks them and
//   publishes ready = it; the others wait for `ready` -- resident_arrive + resident_wait, word for word.
//
// Variants (argument 1):
//   0  SHIPPED: data = relaxed agent-scope atomic stores (sc1), s_waitcnt vmcnt(0), flag = relaxed agent-scope atomic store;
//      reader = relaxed agent-scope polls + s_sleep, compiler barrier, relaxed agent-scope atomic loads
//   1  control: the same WITHOUT the s_waitcnt between data and flag               (flag may overtake the data)
//   2  control: PLAIN data stores / loads + workgroup-scope fence (compiles to nothing on gfx950) (stale per-XCD L2 lines)
//   3  reference: plain data, agent-scope release fence / acquire fence              (the language memory model's way; correct, slow)
//   5  candidate: data in UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached): MTYPE UC, no per-XCD L2 line to
//      go stale), PLAIN 16-byte stores / loads (global_store / load_dwordx4 -- an agent-scope atomic cannot be wider than 8 bytes),
//      s_waitcnt vmcnt(0), relaxed agent-scope flag
//   4  SHIPPED, arrive mode
// Every wait is capped (no hang on a protocol error).  Output: hand-overs, mismatching words, capped waits, ns per hand-over.
//   hipcc --offload-arch=gfx950 -O2 -o mp_litmus mp_litmus.hip && for v in 0 4 1 2 3 5; do timeout 600 ./mp_litmus $v; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void order_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double f(int it, int lane, int q) { return (double)it * 1024.0 + (double)(4 * lane + q); }

__device__ __forceinline__ bool wait_ge(const int* flag, int want, unsigned long long* capped)
{
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 22)) {
            atomicAdd(capped, 1ull);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    return true;
}

typedef double double2v __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(64) k_pairs(int* flags, double* data, int iters, unsigned long long* bad, unsigned long long* capped,
                                              unsigned* xcc_of_block)
{
    const int pair = blockIdx.x >> 1, lane = threadIdx.x;
    const bool writer = (blockIdx.x & 1) == 0;
    int* flag = flags + 32 * (2 * pair);        // (own cache lines)
    int* ack = flags + 32 * (2 * pair + 1);
    double* d = data + 256 * (size_t)pair;
    if (lane == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_of_block[blockIdx.x] = xcc & 0xf;
    }
    unsigned long long nbad = 0;
    for (int it = 1; it <= iters; ++it) {
        if (writer) {
            if (V == 5) {
                double2v* p = reinterpret_cast<double2v*>(d + 4 * lane);
                double2v a = {f(it, lane, 0), f(it, lane, 1)}, b = {f(it, lane, 2), f(it, lane, 3)};
                __builtin_nontemporal_store(a, p);          // (plain 16-byte stores; the memory itself is uncached)
                __builtin_nontemporal_store(b, p + 1);
            } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (V == 2 || V == 3)
                    d[4 * lane + q] = f(it, lane, q);
                else
                    st_agent(d + 4 * lane + q, f(it, lane, q));
            }
            }
            if (V == 0 || V == 5) order_stores();
            if (V == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (V == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            // ONE lane announces the wavefront's stores (like lane 0 of resident_contribute); all 64 lanes' stores are covered by the
            // wavefront-wide s_waitcnt above
            if (lane == 0) __hip_atomic_store(flag, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!wait_ge(ack, it, capped)) return;
        } else {
            if (!wait_ge(flag, it, capped)) return;
            if (V == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (V == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (V == 5) {
                const double2v* p = reinterpret_cast<const double2v*>(d + 4 * lane);
                const double2v a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);
                nbad += (a.x != f(it, lane, 0)) + (a.y != f(it, lane, 1)) + (b.x != f(it, lane, 2)) + (b.y != f(it, lane, 3));
            } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double v = (V == 2 || V == 3) ? ((volatile double*)d)[4 * lane + q] : ld_agent(d + 4 * lane + q);
                nbad += (v != f(it, lane, q));
            }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the reads are done before the writer may overwrite the words
            if (lane == 0) __hip_atomic_store(ack, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// resident_arrive / resident_wait: groups of 4 single-wavefront blocks (4 XCDs); lane 0 of every block is a "branch end"
__global__ void __launch_bounds__(64) k_arrive(int* ints, double* terms, double* result, int iters, unsigned long long* bad,
                                               unsigned long long* capped)
{
    const int grp = blockIdx.x >> 2, me = blockIdx.x & 3;
    if (threadIdx.x != 0) return;
    int* arrive = ints + 32 * (2 * grp);
    int* ready = ints + 32 * (2 * grp + 1);
    unsigned long long nbad = 0;
    for (int it = 1; it <= iters; ++it) {
        double* t = terms + (size_t)(it & 1) * 16 * gridDim.x / 4 + 16 * (size_t)grp + 4 * me;   // terms[it & 1]: ping-pong like rs.terms
#pragma unroll
        for (int q = 0; q < 4; q++) st_agent(t + q, f(it, me, q));
        order_stores();
        const int old = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == 4 * it) {   // the arrival that completes the round: read everybody's terms
            asm volatile("" ::: "memory");
            const double* all = terms + (size_t)(it & 1) * 16 * gridDim.x / 4 + 16 * (size_t)grp;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double v = ld_agent(all + 4 * m + q);
                    nbad += (v != f(it, m, q));
                    s += v;
                }
            st_agent(result + 4 * (size_t)grp + (it & 1), s);
            order_stores();
            __hip_atomic_store(ready, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!wait_ge(ready, it, capped)) return;
        double want = 0.0;
        for (int m = 0; m < 4; m++)
            for (int q = 0; q < 4; q++) want += f(it, m, q);
        nbad += (ld_agent(result + 4 * (size_t)grp + (it & 1)) != want);
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int pairs = argc > 2 ? atoi(argv[2]) : 256;
    const int iters = argc > 3 ? atoi(argv[3]) : 400000;   // 256 x 400 000 = 1.02e8 hand-overs
    int* flags;
    double *data, *result;
    unsigned long long *bad, *capped;
    unsigned* xcc;
    CHECK(hipMalloc(&flags, sizeof(int) * 32 * 2 * (size_t)pairs));
    if (variant == 5)
        CHECK(hipExtMallocWithFlags((void**)&data, sizeof(double) * 256 * (size_t)pairs, hipDeviceMallocUncached));
    else
        CHECK(hipMalloc(&data, sizeof(double) * 256 * (size_t)pairs));
    CHECK(hipMalloc(&result, sizeof(double) * 4 * (size_t)pairs));
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&capped, 8));
    CHECK(hipMalloc(&xcc, sizeof(unsigned) * 4 * (size_t)pairs));
    CHECK(hipMemset(flags, 0, sizeof(int) * 32 * 2 * (size_t)pairs));
    CHECK(hipMemset(data, 0, sizeof(double) * 256 * (size_t)pairs));
    CHECK(hipMemset(result, 0, sizeof(double) * 4 * (size_t)pairs));
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipMemset(capped, 0, 8));
    CHECK(hipMemset(xcc, 0, sizeof(unsigned) * 4 * (size_t)pairs));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    const char* name = "?";
    switch (variant) {
        case 0: name = "SHIPPED (sc1 data, s_waitcnt vmcnt(0), relaxed flag / relaxed poll, sc1 loads)"; hipLaunchKernelGGL(k_pairs<0>, dim3(2 * pairs), dim3(64), 0, 0, flags, data, iters, bad, capped, xcc); break;
        case 1: name = "control: sc1 data, NO s_waitcnt between data and flag"; hipLaunchKernelGGL(k_pairs<1>, dim3(2 * pairs), dim3(64), 0, 0, flags, data, iters, bad, capped, xcc); break;
        case 2: name = "control: plain data + workgroup-scope fence"; hipLaunchKernelGGL(k_pairs<2>, dim3(2 * pairs), dim3(64), 0, 0, flags, data, iters, bad, capped, xcc); break;
        case 5: name = "candidate: UNCACHED memory (hipDeviceMallocUncached), plain 16-byte stores / loads, s_waitcnt, relaxed flag"; hipLaunchKernelGGL(k_pairs<5>, dim3(2 * pairs), dim3(64), 0, 0, flags, data, iters, bad, capped, xcc); break;
        case 3: name = "reference: plain data + agent-scope release / acquire fences"; hipLaunchKernelGGL(k_pairs<3>, dim3(2 * pairs), dim3(64), 0, 0, flags, data, iters, bad, capped, xcc); break;
        default: name = "SHIPPED, arrive mode (4 arrivals per round, the last one reads all terms: resident_arrive)"; hipLaunchKernelGGL(k_arrive, dim3(4 * (pairs / 2)), dim3(64), 0, 0, flags, data, result, iters, bad, capped); break;
    }
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hb = 0, hc = 0;
    CHECK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hc, capped, 8, hipMemcpyDeviceToHost));
    unsigned hx[8] = {0};
    if (variant != 4) CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    const double n = (variant != 4) ? (double)pairs * iters : (double)(pairs / 2) * iters;
    printf("variant %d: %s\n", variant, name);
    printf("  %s x %d rounds = %.3g hand-overs, %.1f ms, %.0f ns per round of one %s\n", variant != 4 ? "pairs" : "groups", iters, n, ms,
           1e6 * ms / iters, variant != 4 ? "pair (there and back)" : "group");
    if (variant != 4) printf("  XCC of blocks 0..7: %u %u %u %u %u %u %u %u (writer / reader of a pair sit on different XCDs)\n", hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7]);
    printf("  mismatching words: %llu    capped waits: %llu\n", hb, hc);
    return 0;
}
