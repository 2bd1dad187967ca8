// Micro-benchmark behind DESIGN.md section 4.1 ("cooperative solver for the stragglers"): what do the dense primitives
// of one TRON step cost when a LONE lane runs them (today's straggler) and when a group of lanes shares them WITHOUT
// changing one rounding (every sum keeps the index order of the lane code)?  Clocks by s_memtime around 256 dependent
// repetitions, one wavefront, one workgroup.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off coop_primitives.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256

__device__ __forceinline__ double bcast(double v, int lane)   // value of `lane` to every lane (lane is a constant)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---- A x, 6 x 6 symmetric (full), lone lane: 36 fused multiply-adds, six independent chains
__global__ void k_symv_lane(const double* in, double* out, unsigned long long* clk)
{
    double A[36], x[6];
    for (int i = 0; i < 36; i++) A[i] = in[i];
    for (int i = 0; i < 6; i++) x[i] = in[36 + i];
    if (threadIdx.x != 0) return;                       // a lone lane, as in the tail of k_branch
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
        double w[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) s = fma(A[6 * i + j], x[j], s);
            w[i] = s;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = w[i] * 0.125;   // next repetition depends on this one
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 6; i++) out[i] = x[i];
    clk[0] = t1 - t0;
}

// ---- the same product shared by six lanes: lane i owns row i (same fma order), results gathered with v_readlane
__global__ void k_symv_group(const double* in, double* out, unsigned long long* clk)
{
    const int lane = threadIdx.x;
    double row[6], x[6];
    for (int j = 0; j < 6; j++) row[j] = in[6 * (lane % 6) + j];
    for (int i = 0; i < 6; i++) x[i] = in[36 + i];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) s = fma(row[j], x[j], s);
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = bcast(s, i) * 0.125;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        for (int i = 0; i < 6; i++) out[i] = x[i];
        clk[0] = t1 - t0;
    }
}

// ---- Cholesky 6 x 6 (left-looking, packed lower), lone lane -- the lane code's chol_inplace
__global__ void k_chol_lane(const double* in, double* out, unsigned long long* clk)
{
    double A0[21];
    for (int i = 0; i < 21; i++) A0[i] = in[64 + i];
    if (threadIdx.x != 0) return;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
        double L[21];
#pragma unroll
        for (int i = 0; i < 21; i++) L[i] = A0[i] + acc;     // depends on the previous factorisation
#pragma unroll
        for (int j = 0; j < 6; j++) {
#pragma unroll
            for (int k = 0; k < j; k++) {
                const double ljk = L[j * (j + 1) / 2 + k];
#pragma unroll
                for (int i = j; i < 6; i++) L[i * (i + 1) / 2 + j] = fma(-L[i * (i + 1) / 2 + k], ljk, L[i * (i + 1) / 2 + j]);
            }
            const double d = sqrt(L[j * (j + 1) / 2 + j]);
            L[j * (j + 1) / 2 + j] = d;
            const double rr = 1.0 / d;
#pragma unroll
            for (int i = j + 1; i < 6; i++) L[i * (i + 1) / 2 + j] *= rr;
        }
        acc = L[20] * 1e-30;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[8] = acc;
    clk[0] = t1 - t0;
}

// ---- the same factorisation with lane i owning row i: same per-entry operation order, pivots / multipliers broadcast
__global__ void k_chol_group(const double* in, double* out, unsigned long long* clk)
{
    const int lane = threadIdx.x, i = lane % 6;
    double r0[6];
    for (int j = 0; j < 6; j++) r0[j] = (j <= i) ? in[64 + i * (i + 1) / 2 + j] : 0.0;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
        double row[6];
#pragma unroll
        for (int j = 0; j < 6; j++) row[j] = r0[j] + acc;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            // entries (i, j), i >= j: subtract L(i,k) L(j,k) for k < j in order; L(j,k) comes from lane j
#pragma unroll
            for (int k = 0; k < j; k++) {
                const double ljk = bcast(row[k], j);
                row[j] = (i >= j) ? fma(-row[k], ljk, row[j]) : row[j];
            }
            const double piv = bcast(row[j], j);
            const double d = sqrt(piv);
            const double rr = 1.0 / d;
            row[j] = (i == j) ? d : ((i > j) ? row[j] * rr : row[j]);
        }
        acc = bcast(row[5], 5) * 1e-30;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        out[9] = acc;
        clk[0] = t1 - t0;
    }
}

// ---- a 6-term dot product: serial chain in one lane vs butterfly over 8 lanes (different rounding!)
__global__ void k_dot_lane(const double* in, double* out, unsigned long long* clk)
{
    double a[6], b[6];
    for (int i = 0; i < 6; i++) { a[i] = in[36 + i]; b[i] = in[42 + i]; }
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int r = 0; r < REP; r++) {
        double t = s * 1e-30;
#pragma unroll
        for (int i = 0; i < 6; i++) t = fma(a[i], b[i], t);
        s = t;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[10] = s;
    clk[0] = t1 - t0;
}
__global__ void k_dot_tree(const double* in, double* out, unsigned long long* clk)
{
    const int lane = threadIdx.x;
    const double a = (lane % 8 < 6) ? in[36 + lane % 8] : 0.0, b = (lane % 8 < 6) ? in[42 + lane % 8] : 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int r = 0; r < REP; r++) {
        double t = fma(a, b, s * 1e-30);
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        s = t;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        out[11] = s;
        clk[0] = t1 - t0;
    }
}

int main()
{
    std::vector<double> h(128);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) h[6 * i + j] = (i == j ? 4.0 : 0.0) + 0.1 / (1 + i + j);
    for (int i = 0; i < 6; i++) { h[36 + i] = 1.0 + 0.1 * i; h[42 + i] = 0.5 - 0.05 * i; }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) h[64 + i * (i + 1) / 2 + j] = h[6 * i + j];
    double *din, *dout;
    unsigned long long* dclk;
    hipMalloc(&din, 128 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dclk, 8);
    hipMemcpy(din, h.data(), 128 * 8, hipMemcpyHostToDevice);
    struct { const char* name; void (*k)(const double*, double*, unsigned long long*); } tests[] = {
        {"A x (6x6)        lone lane, 36 fma                      ", k_symv_lane},
        {"A x (6x6)        6 lanes, row each + 12 v_readlane      ", k_symv_group},
        {"Cholesky 6x6     lone lane                              ", k_chol_lane},
        {"Cholesky 6x6     6 lanes, row each, broadcasts          ", k_chol_group},
        {"dot (6 terms)    lone lane, serial fma chain            ", k_dot_lane},
        {"dot (6 terms)    8 lanes, butterfly (NOT the same bits) ", k_dot_tree},
    };
    printf("# clocks per call (s_memtime, %d dependent repetitions, one wavefront); MI355X\n", REP);
    for (auto& t : tests) {
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; rep++) {
            hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, din, dout, dclk);
            unsigned long long c = 0;
            hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
            if (c < best) best = c;
        }
        printf("%s %8.1f\n", t.name, (double)best / REP);
    }
    return 0;
}
