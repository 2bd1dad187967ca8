// What does a hand-over between two wavefronts through device memory cost INSIDE a kernel on MI355X -- the building block of a
// resident inner loop (DESIGN.md "What comes next" 2b)?  Two single-wavefront blocks play ping-pong:
//     writer: 64 doubles of payload, __threadfence(), flag = it (agent-scope release)
//     reader: spins on the flag (agent-scope acquire loads, s_sleep between polls), reads and checks the payload, answers the same way
// once with the partner on ANOTHER XCD (block 1: blocks go round-robin over the 8 XCDs) and once on the SAME XCD (block 8).
// Also: a chain of dependent agent-scope atomicAdd-with-return from one lane (the per-bus arrival counter).
// Every spin is capped: a protocol error ends the kernel with a failure code instead of hanging the box.
//   hipcc --offload-arch=gfx950 -O2 -o flag_chain flag_chain.hip && timeout 60 ./flag_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ bool wait_flag(int* flag, int it, int* fail)
{
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it) {
        if (++spins > 4000000) {
            *fail = 1;
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__global__ void __launch_bounds__(64) k_pingpong(int* flags, double* payload, int iters, int partner_block, unsigned long long* ticks, int* fail)
{
    const int me = (blockIdx.x == 0) ? 0 : ((int)blockIdx.x == partner_block ? 1 : -1);
    if (me < 0) return;
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (me == 0) {
            payload[lane] = (double)(it + lane);
            __threadfence();
            if (lane == 0) __hip_atomic_store(&flags[0], it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!wait_flag(&flags[1], it, fail)) return;
            if (payload[64 + lane] != (double)(2 * it + lane)) {
                *fail = 2;
                return;
            }
        } else {
            if (!wait_flag(&flags[0], it, fail)) return;
            if (payload[lane] != (double)(it + lane)) {
                *fail = 3;
                return;
            }
            payload[64 + lane] = (double)(2 * it + lane);
            __threadfence();
            if (lane == 0) __hip_atomic_store(&flags[1], it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane == 0) ticks[me] = wall_clock64() - t0;
}

__global__ void __launch_bounds__(64) k_atomic_chain(int* counter, int iters, unsigned long long* ticks, int* sink)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    int v = 0;
    for (int it = 0; it < iters; ++it) v += __hip_atomic_fetch_add(counter + (v & 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticks[0] = wall_clock64() - t0;
    *sink = v;
}

int main()
{
    int *flags, *fail, *counter, *sink;
    double* payload;
    unsigned long long* ticks;
    CHECK(hipMalloc(&flags, 2 * sizeof(int)));
    CHECK(hipMalloc(&fail, sizeof(int)));
    CHECK(hipMalloc(&counter, 2 * sizeof(int)));
    CHECK(hipMalloc(&sink, sizeof(int)));
    CHECK(hipMalloc(&payload, 128 * sizeof(double)));
    CHECK(hipMalloc(&ticks, 2 * sizeof(unsigned long long)));
    int rate_khz = 0;
    CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double ns_per_tick = 1e6 / rate_khz;
    const int iters = 2000;
    for (int partner : {1, 8, 2, 9}) {
        CHECK(hipMemset(flags, 0, 2 * sizeof(int)));
        CHECK(hipMemset(fail, 0, sizeof(int)));
        hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(64), 0, 0, flags, payload, iters, partner, ticks, fail);
        CHECK(hipDeviceSynchronize());
        unsigned long long t[2];
        int f;
        CHECK(hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&f, fail, sizeof(int), hipMemcpyDeviceToHost));
        printf("ping-pong with block %d (%s XCD): %d round trips, fail code %d, %.0f ns per round trip = %.0f ns per hand-over (payload + fence + flag -> acquire + read)\n",
               partner, (partner % 8 == 0) ? "same" : "other", iters, f, t[0] * ns_per_tick / iters, t[0] * ns_per_tick / iters / 2);
    }
    CHECK(hipMemset(counter, 0, 2 * sizeof(int)));
    hipLaunchKernelGGL(k_atomic_chain, dim3(1), dim3(64), 0, 0, counter, 4000, ticks, sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long t0;
    CHECK(hipMemcpy(&t0, ticks, sizeof(t0), hipMemcpyDeviceToHost));
    printf("dependent agent-scope atomicAdd with return, one lane: %.0f ns each\n", t0 * ns_per_tick / 4000);
    return 0;
}
