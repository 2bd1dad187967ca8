// TEST INFRASTRUCTURE ONLY -- occupies `ncu` compute units of the GPU for `seconds`: one block per CU that takes (almost) all of
// the CU's LDS, so that no block of another process's kernel that needs LDS can be placed there.  tests/test_gpu_resident.py runs
// it beside a resident loop: with most CUs taken, the loop's wavefronts cannot all be on the chip at once, its waits run into
// their cap, and the library must fall back to the two-launch loop by itself.
//   hipcc --offload-arch=gfx950 -O2 -o gpu_hog gpu_hog.hip && ./gpu_hog 200 3.0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(64) k_hog(unsigned long long ticks, int* sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += lds[(threadIdx.x + acc) & 63];
        __builtin_amdgcn_s_sleep(64);
    }
    if (acc == -1) *sink = acc;
}

int main(int argc, char** argv)
{
    const int ncu = argc > 1 ? atoi(argv[1]) : 200;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    const int lds_bytes = 150 * 1024;   // of 160 KB per CU: the second block of anything with >= 10 KB of LDS does not fit beside it
    if (hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        fprintf(stderr, "cannot raise the dynamic LDS limit\n");
        return 2;
    }
    int* sink = nullptr;
    if (hipMalloc(&sink, 4) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_hog, dim3(ncu), dim3(64), lds_bytes, 0, (unsigned long long)(seconds * 1e8), sink);   // wall_clock64: 100 MHz
    printf("HOG RUNNING\n");
    fflush(stdout);
    const hipError_t e = hipDeviceSynchronize();
    printf("HOG DONE %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
