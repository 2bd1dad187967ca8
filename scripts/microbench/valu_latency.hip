// Microbenchmark (gfx950): what does ONE wavefront alone on a SIMD pay per FP64 instruction -- as a dependent chain and
// as 2 / 4 independent chains -- and what do the instruction kinds cost that the branch solver's lone-lane trips are
// made of (fma, mul+add, compare+select, min/max, division, square root, AGPR round trip, readlane/writelane)?
// The branch kernel runs one wave per SIMD and its launch time is the serial chain of its slowest lane, so these numbers
// say how much of a trip is latency (shortened by instruction-level parallelism) and how much is issue.
//   hipcc --offload-arch=gfx950 -O2 -o valu_latency valu_latency.hip && ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define N_OUTER 64

__global__ void k_lat(double* out, unsigned long long* cyc, double a0, double b0)
{
    double a = a0, b = b0;
    double x0 = a0 + threadIdx.x, x1 = x0 + 1.0, x2 = x0 + 2.0, x3 = x0 + 3.0;
    unsigned long long t0, t1;
    int q = 0;
#define TIME_BEGIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory")
#define TIME_END(n_instr)                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory"); \
    if (threadIdx.x == 0) cyc[q] = (t1 - t0);                                                                               \
    q++;

    // 0: dependent v_fma_f64 chain, 1024 instructions
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER; i++) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
    TIME_END(1024)
    // 1: two independent chains, 1024 instructions in all
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER / 2; i++) {
        REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));)
    }
    TIME_END(1024)
    // 2: four independent chains
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER / 4; i++) {
        REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                           : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));)
    }
    TIME_END(1024)
    // 3: dependent v_mul_f64 + v_add_f64 pairs (the unfused form), 1024 instructions
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER / 2; i++) { REP16(asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2" : "+v"(x0) : "v"(a), "v"(b));) }
    TIME_END(1024)
    // 4: dependent compare + select chain: x = (x < b) ? x + a : x   -> v_cmp_lt_f64, v_add_f64, 2 x v_cndmask_b32 (4 instr per step, 256 steps)
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER / 4; i++) {
        REP16(x0 = (x0 < b) ? x0 + a : x0; asm volatile("" : "+v"(x0));)
    }
    TIME_END(1024)
    // 5: dependent v_max_f64 / v_min_f64 chain (clamp), 1024 instructions
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER / 2; i++) { REP16(asm volatile("v_max_f64 %0, %0, %1\n\tv_min_f64 %0, %0, %2" : "+v"(x0) : "v"(a), "v"(b));) }
    TIME_END(1024)
    // 6: dependent f32 fma chain for reference
    {
        float f = (float)x0, fa = (float)a, fb = (float)b;
        TIME_BEGIN();
        for (int i = 0; i < N_OUTER; i++) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(fa), "v"(fb));) }
        x1 += f;
        TIME_END(1024)
    }
    // 7: AGPR round trip chain: v_accvgpr_write + v_accvgpr_read (both halves of a double = 4 instructions per step), 256 steps
    TIME_BEGIN();
    {
        int v = threadIdx.x;
        for (int i = 0; i < N_OUTER / 2; i++) { REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(v) : : "a0");) }
        x3 += v;
    }
    TIME_END(1024)
    // 8: v_writelane / v_readlane round trip through an SGPR (SGPR spill traffic): 2 instructions per step, 512 steps
    {
        int v = threadIdx.x, s;
        TIME_BEGIN();
        for (int i = 0; i < N_OUTER / 2; i++) { REP16(asm volatile("v_readlane_b32 %1, %0, 3\n\tv_writelane_b32 %0, %1, 5" : "+v"(v), "=s"(s));) }
        x2 += v;
        TIME_END(1024)
    }
    // 9: double division chain (compiler's IEEE sequence), 64 divisions
    TIME_BEGIN();
    for (int i = 0; i < 64; i++) {
        x0 = b / x0;
        asm volatile("" : "+v"(x0));
    }
    TIME_END(64)
    // 10: double square root chain, 64
    TIME_BEGIN();
    for (int i = 0; i < 64; i++) {
        x0 = sqrt(x0 + a);
        asm volatile("" : "+v"(x0));
    }
    TIME_END(64)
    // 11: sincos, 16
    TIME_BEGIN();
    for (int i = 0; i < 16; i++) {
        double sn, cs;
        sincos(x0, &sn, &cs);
        x0 = sn + cs;
        asm volatile("" : "+v"(x0));
    }
    TIME_END(16)
    // 12: four INDEPENDENT divisions per step, 16 steps (does the compiler's sequence overlap?)
    TIME_BEGIN();
    for (int i = 0; i < 16; i++) {
        x0 = b / x0; x1 = b / x1; x2 = b / x2; x3 = b / x3;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    }
    TIME_END(64)
    // 13: s_and_saveexec + s_mov exec round trip (divergent-if bookkeeping), 1024 pairs -- SALU only
    TIME_BEGIN();
    for (int i = 0; i < N_OUTER; i++) { REP16(asm volatile("s_and_saveexec_b64 s[20:21], vcc\n\ts_mov_b64 exec, s[20:21]" : : : "s20", "s21", "scc");) }
    TIME_END(2048)
    // 14: VALU -> SGPR -> VALU dependency: v_cmp writes vcc, s_and_b64 consumes it, v_cndmask consumes the SALU result (3 per step, 256 steps)
    TIME_BEGIN();
    {
        float f = (float)x0, fb = (float)b;
        for (int i = 0; i < N_OUTER / 4; i++) {
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(fb) : "vcc", "scc");)
        }
        x1 += f;
    }
    TIME_END(768)
    out[threadIdx.x] = x0 + x1 + x2 + x3;
}

int main()
{
    double* d_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_out, 64 * sizeof(double));
    (void)hipMalloc(&d_cyc, 32 * sizeof(unsigned long long));
    (void)hipMemset(d_cyc, 0, 32 * sizeof(unsigned long long));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 0.999999, 1.0000001);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> c(32);
    (void)hipMemcpy(c.data(), d_cyc, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz: convert with the shader clock (2.4 GHz measured on this part, profiles/archive_r01_r02/r02_pmc_summary.json)
    const double f = 24.0;
    const char* name[] = {"v_fma_f64 dependent chain", "v_fma_f64 2 independent chains", "v_fma_f64 4 independent chains",
                          "v_mul_f64 + v_add_f64 dependent", "cmp + add + 2 cndmask (select) dependent", "v_max_f64 + v_min_f64 dependent",
                          "v_fma_f32 dependent chain", "v_accvgpr_write + v_accvgpr_read (per pair)", "v_readlane + v_writelane",
                          "double division (per division)", "double sqrt (+1 add, per sqrt)", "sincos (per call)",
                          "4 independent divisions (per division)", "s_and_saveexec + s_mov exec (per pair)", "v_cmp -> s_and -> v_cndmask (per 3)"};
    const int n[] = {1024, 1024, 1024, 1024, 256, 1024, 1024, 512, 512, 64, 64, 16, 64, 1024, 256};
    printf("one wavefront alone on its SIMD, s_memtime ticks x %.0f (2.4 GHz shader clock / 100 MHz counter)\n", f);
    for (int q = 0; q < 15; q++) printf("%-48s %8.1f clocks per unit  (%llu ticks / %d)\n", name[q], (double)c[q] * f / n[q], c[q], n[q]);
    return 0;
}
