// What does a dependent kernel launch cost on this GPU when the kernel does (almost) nothing -- and what makes it dearer?
// hipcc --offload-arch=gfx950 -O2 launch_floor.hip -o launch_floor && ./launch_floor
// Each variant is launched 400 times back to back on one stream (every launch depends on the previous one, as the two
// launches of an ADMM iteration do); the figure is stream time per launch (HIP events around the whole sequence).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { void* p[80]; };   // the size of the library's Dev block (passed by value to the bus kernel)

__global__ void k_empty() {}
__global__ void k_flag(const int* flag, int* out) { if (*flag) out[0] = 1; }
__global__ void k_flag_big(Big b, const int* flag, int* out) { if (*flag) out[0] = (int)(size_t)b.p[79]; }
__global__ void __launch_bounds__(256) k_flag_lds(const int* flag, int* out)
{
    __shared__ double s[2368];   // 18.9 KB
    if (*flag) { s[threadIdx.x] = 1.0; __syncthreads(); out[0] = (int)s[(threadIdx.x + 1) % 256]; }
}
// one wave per SIMD register footprint: touch a high VGPR and an AGPR so that the kernel descriptor asks for 512
__global__ void __launch_bounds__(64, 1) k_flag_regs(const int* flag, int* out)
{
    if (*flag) {
        asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a200, v255" ::: "v255", "a200");
        out[0] = 1;
    }
}
// the flag was written by the previous launch (as the loop-control word is): read-after-remote-write
__global__ void k_flag_pingpong(int* flag, int* out) { int f = *flag; if (f < 0) out[0] = 1; if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *flag = f + 1; }
__global__ void k_touch(double* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * 1.0000001 + 1.0; }

template <class F>
static double time_launches(F launch, int reps = 400)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / reps;
}

int main()
{
    int *flag, *out;
    double* a;
    hipMalloc(&flag, 4); hipMalloc(&out, 4); hipMalloc(&a, 8 << 20);
    hipMemset(flag, 0, 4); hipMemset(a, 0, 8 << 20);
    Big big{};
    printf("%-44s", "workgroups:");
    const int grids[] = {1, 64, 384, 428, 1024, 4096};
    for (int g : grids) printf("%8d", g);
    printf("   (us per launch)\n");
#define ROW(name, block, call)                                  \
    printf("%-44s", name);                                      \
    for (int g : grids) { (void)g; printf("%8.2f", time_launches([&] { call; })); } \
    printf("\n");
    ROW("empty, 64 threads", 64, hipLaunchKernelGGL(k_empty, dim3(g), dim3(64), 0, 0))
    ROW("empty, 256 threads", 256, hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, 0))
    ROW("one flag load, 64 threads", 64, hipLaunchKernelGGL(k_flag, dim3(g), dim3(64), 0, 0, flag, out))
    ROW("one flag load, 256 threads", 256, hipLaunchKernelGGL(k_flag, dim3(g), dim3(256), 0, 0, flag, out))
    ROW("flag load + 640 B by-value argument, 256 thr", 256, hipLaunchKernelGGL(k_flag_big, dim3(g), dim3(256), 0, 0, big, flag, out))
    ROW("flag load + 18.9 KB LDS per block, 256 thr", 256, hipLaunchKernelGGL(k_flag_lds, dim3(g), dim3(256), 0, 0, flag, out))
    ROW("flag load + 512-register waves, 64 thr", 64, hipLaunchKernelGGL(k_flag_regs, dim3(g), dim3(64), 0, 0, flag, out))
    ROW("flag written by the previous launch, 64 thr", 64, hipLaunchKernelGGL(k_flag_pingpong, dim3(g), dim3(64), 0, 0, flag, out))
    ROW("1 read + 1 write per thread, 256 thr", 256, hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, 0, a, g * 256))
    return 0;
}
