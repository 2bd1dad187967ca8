// Which streams of ONE process can hold each other up?  stream a: spin(flag) [+ something queued behind it]; 2 ms later, stream b: set(flag).  If
// set() only starts when spin() has run into its cap, b's kernel sat behind a's: the two streams share a hardware queue (the runtime maps a process's
// streams onto a pool of GPU_MAX_HW_QUEUES queues, 4 by default).  What is queued behind the spinning kernel makes no difference, and with a queue per
// stream nothing is held up (MI355X, ROCm 7: profiles/r06_heavy_split_on_partitions.txt).  Why it matters here: the tests' ranks are THREADS of one
// process on one GPU, and their resident kernels / sums exchanges wait for each other -- two ranks on one queue is a deadlock until a wait cap ends it.
// build: hipcc -O2 --offload-arch=gfx950 queue_pairs.hip -o queue_pairs;  run: GPU_MAX_HW_QUEUES=4 ./queue_pairs 10 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include <chrono>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(int* flag, long long cap, long long* out)
{
    const long long t0 = wall_clock64();
    int seen = 0;
    while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) && wall_clock64() - t0 < cap) __builtin_amdgcn_s_sleep(32);
    out[0] = t0; out[1] = wall_clock64(); out[2] = seen;
}
__global__ void setflag(int* flag, long long* out) { out[0] = wall_clock64(); __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void trailing(long long* out) { out[0] = wall_clock64(); }
int main(int argc, char** argv)
{
    const int ns = argc > 1 ? atoi(argv[1]) : 8;
    const int mode = argc > 2 ? atoi(argv[2]) : 1;   // 0: nothing behind spin; 1: a kernel behind it; 2: an async D2H copy behind it; 3: an event record behind it
    const long long cap = 5000000;   // 50 ms
    std::vector<hipStream_t> st(ns);
    for (auto& s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* flag; long long *o1, *o2, *o3, *h;
    CHK(hipExtMallocWithFlags((void**)&flag, 64, hipDeviceMallocFinegrained));
    CHK(hipMalloc(&o1, 64)); CHK(hipMalloc(&o2, 64)); CHK(hipMalloc(&o3, 64)); CHK(hipHostMalloc(&h, 64));
    hipEvent_t ev; CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto& s : st) { hipLaunchKernelGGL(trailing, dim3(1), dim3(64), 0, s, o3); }   // (every stream used once: its queue exists)
    CHK(hipDeviceSynchronize());
    int blocked = 0, total = 0;
    printf("streams %d, mode %d (what is queued behind the spinning kernel: 0 nothing, 1 kernel, 2 async copy, 3 event record)\nrow = spinning stream, column = stream of the kernel that sets the flag; . flag seen in time, X only after the cap\n", ns, mode);
    for (int a = 0; a < ns; a++) {
        for (int b = 0; b < ns; b++) {
            if (a == b) { printf(" -"); continue; }
            CHK(hipMemset(flag, 0, 64)); CHK(hipDeviceSynchronize());
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[a], flag, cap, o1);
            if (mode == 1) hipLaunchKernelGGL(trailing, dim3(1), dim3(64), 0, st[a], o3);
            if (mode == 2) CHK(hipMemcpyAsync(h, o1, 32, hipMemcpyDeviceToHost, st[a]));
            if (mode == 3) CHK(hipEventRecord(ev, st[a]));
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            hipLaunchKernelGGL(setflag, dim3(1), dim3(64), 0, st[b], flag, o2);
            CHK(hipStreamSynchronize(st[a])); CHK(hipStreamSynchronize(st[b]));
            long long r1[3], r2[1];
            CHK(hipMemcpy(r1, o1, 24, hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2, o2, 8, hipMemcpyDeviceToHost));
            const bool x = r1[2] == 0;
            printf(" %c", x ? 'X' : '.');
            blocked += x; total++;
        }
        printf("\n");
    }
    printf("blocked pairs: %d of %d\n", blocked, total);
    return 0;
}
