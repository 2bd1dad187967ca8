// What does a dependency BETWEEN two HIP streams cost on MI355X (hipEventRecord on one stream, hipStreamWaitEvent on the other),
// compared with the implicit dependency between two kernels of ONE stream?  The multi-GPU loop (dist_native.h
// enqueue_dist_iterations) has two such hops per iteration on the path bus launch -> exchange -> cut branches -> bus launch.
//   chain A: 2 n tiny kernels on one stream
//   chain B: n x (kernel on M, record, X waits, kernel on X, record, M waits)        -- direct enqueue
//   chain C: chain B captured into a hipGraph and replayed
//   chain D: like B, with a long kernel (~50 us) running on M beside the X kernel    -- is the X chain slowed down by work on M?
//   hipcc --offload-arch=gfx950 -O2 -o stream_hop stream_hop.hip && ./stream_hop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_tiny(int* p) { if (threadIdx.x == 0 && p) atomicAdd(p, 1); }
__global__ void k_long(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16); }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t M, X;
    CHECK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&X, hipStreamNonBlocking));
    int* d;
    CHECK(hipMalloc(&d, 4));
    const int n = 200;
    hipEvent_t ev[2 * n + 2];
    for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // chain E: per round a 50-us kernel on EACH stream (concurrent: ~50 us per round; serialised: ~100)
    auto chainE = [&]() {
        CHECK(hipEventRecord(ev[0], M));
        CHECK(hipStreamWaitEvent(X, ev[0], 0));
        for (int i = 0; i < n; i++) {
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, M, d);
            CHECK(hipEventRecord(ev[2 + 2 * i], M));
            hipLaunchKernelGGL(k_long, dim3(256), dim3(64), 0, M, 5000ull);
            CHECK(hipStreamWaitEvent(X, ev[2 + 2 * i], 0));
            hipLaunchKernelGGL(k_long, dim3(64), dim3(64), 0, X, 5000ull);
            CHECK(hipEventRecord(ev[3 + 2 * i], X));
            CHECK(hipStreamWaitEvent(M, ev[3 + 2 * i], 0));
        }
        CHECK(hipEventRecord(ev[1], X));
        CHECK(hipStreamWaitEvent(M, ev[1], 0));
    };
    auto chainB = [&](bool with_long) {
        CHECK(hipEventRecord(ev[0], M));
        CHECK(hipStreamWaitEvent(X, ev[0], 0));
        for (int i = 0; i < n; i++) {
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, M, d);
            CHECK(hipEventRecord(ev[2 + 2 * i], M));
            if (with_long) hipLaunchKernelGGL(k_long, dim3(256), dim3(64), 0, M, 5000ull);   // 50 us at 100 MHz
            CHECK(hipStreamWaitEvent(X, ev[2 + 2 * i], 0));
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, X, d);
            CHECK(hipEventRecord(ev[3 + 2 * i], X));
            CHECK(hipStreamWaitEvent(M, ev[3 + 2 * i], 0));
        }
        CHECK(hipEventRecord(ev[1], X));
        CHECK(hipStreamWaitEvent(M, ev[1], 0));
    };
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < 2 * n; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, M, d);
        CHECK(hipStreamSynchronize(M));
        const double a = now() - t0;
        t0 = now();
        chainB(false);
        CHECK(hipStreamSynchronize(M));
        const double b = now() - t0;
        t0 = now();
        chainB(true);
        CHECK(hipStreamSynchronize(M));
        const double dd = now() - t0;
        printf("rep %d: A one stream %.2f us per kernel | B two streams, direct %.2f us per kernel (incl. one hop) | D with a 50-us kernel on M per round: %.2f us per round\n",
               rep, 1e6 * a / (2 * n), 1e6 * b / (2 * n), 1e6 * dd / n);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(M, hipStreamCaptureModeThreadLocal));
    chainB(false);
    CHECK(hipStreamEndCapture(M, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipGraph_t g2;
    hipGraphExec_t ge2;
    CHECK(hipStreamBeginCapture(M, hipStreamCaptureModeThreadLocal));
    chainB(true);
    CHECK(hipStreamEndCapture(M, &g2));
    CHECK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    hipGraph_t g3;
    hipGraphExec_t ge3;
    CHECK(hipStreamBeginCapture(M, hipStreamCaptureModeThreadLocal));
    chainE();
    CHECK(hipStreamEndCapture(M, &g3));
    CHECK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipDeviceSynchronize());
        double t0 = now();
        chainE();
        CHECK(hipStreamSynchronize(M));
        const double e1 = now() - t0;
        t0 = now();
        CHECK(hipGraphLaunch(ge3, M));
        CHECK(hipStreamSynchronize(M));
        const double e2 = now() - t0;
        printf("rep %d: E a 50-us kernel on EACH stream per round: direct %.2f us per round | hipGraph %.2f us per round  (concurrent ~50, serialised ~100)\n", rep, 1e6 * e1 / n, 1e6 * e2 / n);
    }
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipDeviceSynchronize());
        double t0 = now();
        CHECK(hipGraphLaunch(ge, M));
        CHECK(hipStreamSynchronize(M));
        const double c = now() - t0;
        t0 = now();
        CHECK(hipGraphLaunch(ge2, M));
        CHECK(hipStreamSynchronize(M));
        const double c2 = now() - t0;
        printf("rep %d: C two streams, hipGraph %.2f us per kernel (incl. one hop) | with the 50-us kernel: %.2f us per round\n", rep, 1e6 * c / (2 * n), 1e6 * c2 / n);
    }
    return 0;
}
