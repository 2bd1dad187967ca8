// What does the TCC FETCH_SIZE counter report for the access shapes of this library?  (MI355X_MICROARCH.md: FETCH_SIZE is
// in KB and counts a coalesced 128-B request as 64 B -- documented for 16-B-per-lane streams.  The bus update reads with
// 8-B gathers and, since round 3, 64-B records; VERDICT r02 asked for the correction to be checked on known byte counts.)
// Three kernels over a 1 GiB buffer (4x the Infinity Cache, every byte read once, so HBM traffic = bytes touched):
//   k_stream16 : 16 B per lane, contiguous       -> N bytes requested, N bytes touched
//   k_stream8  :  8 B per lane, contiguous       -> N bytes requested, N bytes touched
//   k_gather8  :  8 B per lane at the start of a DISTINCT random 128-B line -> 8 B requested per lane, one line touched
//   k_rec64    : 64 B per lane (8 doubles), records at random 64-B-aligned places -> 64 B requested and touched per lane
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_size_check
// and compare Counter_Value x 1024 with the byte counts this program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

__global__ void k_stream16(const double2* p, size_t n, double* out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 1.2345) out[0] = s;
}
__global__ void k_stream8(const double* p, size_t n, double* out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 1.2345) out[0] = s;
}
__global__ void k_gather8(const double* p, const unsigned* line, size_t nlines, double* out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) s += p[(size_t)line[i] * 16];
    if (s == 1.2345) out[0] = s;
}
__global__ void k_rec64(const double* p, const unsigned* slot, size_t nslots, double* out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
        const double* q = p + (size_t)slot[i] * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) s += q[k];
    }
    if (s == 1.2345) out[0] = s;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    double *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, bytes);
    const size_t nlines = bytes / 128 / 4;   // a quarter of the lines, each exactly once
    std::vector<unsigned> line(bytes / 128);
    std::iota(line.begin(), line.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(line.begin(), line.end(), rng);
    line.resize(nlines);
    const size_t nslots = bytes / 64 / 4;
    std::vector<unsigned> slot(bytes / 64);
    std::iota(slot.begin(), slot.end(), 0u);
    std::shuffle(slot.begin(), slot.end(), rng);
    slot.resize(nslots);
    unsigned *dline, *dslot;
    (void)hipMalloc(&dline, nlines * 4);
    (void)hipMalloc(&dslot, nslots * 4);
    (void)hipMemcpy(dline, line.data(), nlines * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dslot, slot.data(), nslots * 4, hipMemcpyHostToDevice);
    const int nb = 256 * 8;
    hipLaunchKernelGGL(k_stream16, dim3(nb), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_stream8, dim3(nb), dim3(256), 0, 0, buf, bytes / 8, out);
    hipLaunchKernelGGL(k_gather8, dim3(nb), dim3(256), 0, 0, buf, dline, nlines, out);
    hipLaunchKernelGGL(k_rec64, dim3(nb), dim3(256), 0, 0, buf, dslot, nslots, out);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("k_stream16 : requested %zu B, touched %zu B\n", bytes, bytes);
    printf("k_stream8  : requested %zu B, touched %zu B\n", bytes, bytes);
    printf("k_gather8  : requested %zu B (+ %zu B of indices), distinct 128-B lines touched %zu = %zu B (%zu B as 64-B sectors)\n", nlines * 8, nlines * 4,
           nlines, nlines * 128, nlines * 64);
    printf("k_rec64    : requested %zu B (+ %zu B of indices), touched %zu B\n", nslots * 64, nslots * 4, nslots * 64);
    return 0;
}
