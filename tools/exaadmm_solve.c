/*
 * exaadmm_solve -- command-line driver over the C ABI (include/exaadmm_hip.h), plain C.
 * What `julia -e 'ExaAdmm.solve_acopf(case; ...)'` does with the reference (interface/solve_acopf.jl:1-46,
 * utils/print_statistics.jl:1-21): load a MATPOWER case with the library's native loader, build the device
 * model, run the two-level ADMM on one MI355X, print the statistics block.
 *
 *   exaadmm_solve case.m [--rho_pq R] [--rho_va R] [--scale S] [--tight_factor F] [--outer N] [--inner N]
 *                        [--outer_eps E] [--gpu K] [--pglib] [--quiet]
 *                        [--load_prefix P --start_period A --end_period B [--ramp_ratio R]]
 *
 * With --load_prefix the multi-period model is solved (interface/solve_mpacopf.jl with warm_start=false):
 * P.Pd / P.Qd are whitespace matrices nbus x T (utils/opfdata.jl:121-130), periods A..B (1-based) are used.
 * With --rolling in addition, the periods are solved one after another instead (interface/solve_acopf_rolling.jl,
 * models/acopf/acopf_admm_rolling_cpu.jl:16-48): load swap, warm start from the previous period, generation
 * bounds limited to +-ramp_ratio*pgmax around the previous dispatch.
 *
 * Exit code 0 = solved, 1 = iteration limit, 2 = error (message on stderr).
 */
#include "exaadmm_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reads an nbus x T whitespace matrix; returns period-major T x nbus (columns one after another) */
static double *read_load(const char *path, long nbus, int *T)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    size_t cap = 1024, n = 0;
    double *v = (double *)malloc(cap * sizeof(double)), x;
    while (fscanf(f, "%lf", &x) == 1) {
        if (n == cap) v = (double *)realloc(v, (cap *= 2) * sizeof(double));
        v[n++] = x;
    }
    fclose(f);
    if (nbus <= 0 || n == 0 || n % (size_t)nbus != 0) {
        free(v);
        return NULL;
    }
    *T = (int)(n / (size_t)nbus);
    double *out = (double *)malloc(n * sizeof(double));
    for (long b = 0; b < nbus; b++)
        for (int t = 0; t < *T; t++) out[(size_t)t * nbus + b] = v[(size_t)b * *T + t];
    free(v);
    return out;
}

static int fail(const char *what)
{
    const char *m = exaadmm_last_error();
    if (!m || !*m) m = exaadmm_io_error();
    fprintf(stderr, "exaadmm_solve: %s: %s\n", what, m ? m : "?");
    return 2;
}

int main(int argc, char **argv)
{
    if (argc < 2 || strcmp(argv[1], "-h") == 0 || strcmp(argv[1], "--help") == 0) {
        fprintf(stderr, "usage: %s case.m [--rho_pq R] [--rho_va R] [--scale S] [--tight_factor F] [--outer N] "
                        "[--inner N] [--outer_eps E] [--gpu K] [--pglib] [--quiet]\n(%s)\n", argv[0], exaadmm_version());
        return 2;
    }
    const char *path = argv[1];
    double rho_pq = 4e2, rho_va = 4e4, tight = 1.0, ramp_ratio = 0.02;
    int gpu = 0, quiet = 0, p0 = 1, p1 = 1, rolling = 0;
    const char *load_prefix = NULL;
    const char *fmt = "matpower";
    exaadmm_params par;
    exaadmm_default_params(&par);
    for (int i = 2; i < argc; i++) {
        const char *a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
#define OPT(name) (strcmp(a, name) == 0 && v && ++i)
        if (OPT("--rho_pq")) rho_pq = atof(v);
        else if (OPT("--rho_va")) rho_va = atof(v);
        else if (OPT("--scale")) par.scale = atof(v);
        else if (OPT("--tight_factor")) tight = atof(v);
        else if (OPT("--outer")) par.outer_iterlim = atoi(v);
        else if (OPT("--inner")) par.inner_iterlim = atoi(v);
        else if (OPT("--outer_eps")) par.outer_eps = atof(v);
        else if (OPT("--gpu")) gpu = atoi(v);
        else if (OPT("--load_prefix")) load_prefix = v;
        else if (OPT("--start_period")) p0 = atoi(v);
        else if (OPT("--end_period")) p1 = atoi(v);
        else if (OPT("--ramp_ratio")) ramp_ratio = atof(v);
        else if (strcmp(a, "--pglib") == 0) fmt = "pglib";
        else if (strcmp(a, "--quiet") == 0) quiet = 1;
        else if (strcmp(a, "--rolling") == 0) rolling = 1;
        else {
            fprintf(stderr, "exaadmm_solve: unknown or incomplete option %s\n", a);
            return 2;
        }
#undef OPT
    }
    par.verbose = quiet ? 0 : 1;

    exaadmm_case *cs = NULL;
    if (exaadmm_load_matpower(path, tight, fmt, &cs) != EXAADMM_OK) return fail("load");
    exaadmm_grid grid;
    if (exaadmm_case_grid(cs, &grid) != EXAADMM_OK) return fail("grid");
    exaadmm_handle *h = NULL;
    int rc_create;
    double *pd = NULL, *qd = NULL;
    if (load_prefix) {
        char fn[4096];
        int T1 = 0, T2 = 0;
        snprintf(fn, sizeof(fn), "%s.Pd", load_prefix);
        pd = read_load(fn, (long)grid.nbus, &T1);
        snprintf(fn, sizeof(fn), "%s.Qd", load_prefix);
        qd = read_load(fn, (long)grid.nbus, &T2);
        if (!pd || !qd || T1 != T2 || p0 < 1 || p1 < p0 || p1 > T1) {
            fprintf(stderr, "exaadmm_solve: load files %s.Pd/.Qd missing, not nbus x T, or periods out of range\n", load_prefix);
            free(pd);
            free(qd);
            exaadmm_case_free(cs);
            return 2;
        }
    } else if (rolling) {
        fprintf(stderr, "exaadmm_solve: --rolling needs --load_prefix\n");
        exaadmm_case_free(cs);
        return 2;
    }
    if (load_prefix && !rolling)
        rc_create = exaadmm_mp_create(&grid, &par, p1 - p0 + 1, pd + (size_t)(p0 - 1) * grid.nbus, qd + (size_t)(p0 - 1) * grid.nbus,
                                      ramp_ratio, gpu, &h);
    else
        rc_create = exaadmm_create(&grid, &par, gpu, &h);
    if (rc_create != EXAADMM_OK) {
        free(pd);
        free(qd);
        exaadmm_case_free(cs);
        return fail("create");
    }
    if (rolling) {
        /* admm_restart_rolling (acopf_admm_rolling_cpu.jl:16-48) */
        const long G = (long)grid.ngen, nvar = (long)exaadmm_nvar(h);
        double *u = (double *)malloc((size_t)nvar * sizeof(double));
        double *lo = (double *)malloc((size_t)(G > 0 ? G : 1) * sizeof(double)), *hi = (double *)malloc((size_t)(G > 0 ? G : 1) * sizeof(double));
        int rc = exaadmm_init_solution(h, rho_pq, rho_va), all_solved = 1;
        exaadmm_info info;
        for (int t = p0; rc == EXAADMM_OK && t <= p1; t++) {
            rc = exaadmm_set_load(h, pd + (size_t)(t - 1) * grid.nbus, qd + (size_t)(t - 1) * grid.nbus, grid.nbus);
            if (rc == EXAADMM_OK) rc = exaadmm_solve_two_level(h, &info);
            if (rc != EXAADMM_OK) break;
            printf(" ** Statistics of time period %d\n", t);
            printf("Status  . . . . . . . . . . . . . . . . . %s\n", info.status == 1 ? "Solved" : "IterationLimit");
            printf("Objective value . . . . . . . . . . . . . %.6e\n", info.objval);
            printf("Residual  . . . . . . . . . . . . . . . . %.6e\n", info.mismatch);
            printf("Cumulative iterations . . . . . . . . . . %d\n", info.cumul);
            printf("Time (secs) . . . . . . . . . . . . . . . %5.3f\n", info.time_overall);
            all_solved = all_solved && info.status == 1;
            rc = exaadmm_download_solution(h, EXAADMM_U_CURR, u, nvar);
            for (long g = 0; g < G; g++) {   /* update_real_power_current_bounds (:1-14) */
                const double r = ramp_ratio * grid.pgmax[g];
                lo[g] = grid.pgmin[g] > u[2 * g] - r ? grid.pgmin[g] : u[2 * g] - r;
                hi[g] = grid.pgmax[g] < u[2 * g] + r ? grid.pgmax[g] : u[2 * g] + r;
            }
            if (rc == EXAADMM_OK && G > 0) rc = exaadmm_set_pg_bounds(h, lo, hi, G);
        }
        free(u); free(lo); free(hi); free(pd); free(qd);
        if (rc != EXAADMM_OK) fail("rolling solve");
        exaadmm_destroy(h);
        exaadmm_case_free(cs);
        return rc != EXAADMM_OK ? 2 : (all_solved ? 0 : 1);
    }
    free(pd);
    free(qd);
    int rc = exaadmm_init_solution(h, rho_pq, rho_va);
    exaadmm_info info;
    memset(&info, 0, sizeof(info));
    if (rc == EXAADMM_OK) rc = exaadmm_solve_two_level(h, &info);
    if (rc != EXAADMM_OK) {
        fail("solve");
        exaadmm_destroy(h);
        exaadmm_case_free(cs);
        return 2;
    }
    /* print_statistics (utils/print_statistics.jl:1-21) */
    printf(" ** Statistics of %s (%lld generators, %lld branches, %lld buses, %d period%s)\n", path, (long long)grid.ngen,
           (long long)grid.nline, (long long)grid.nbus, exaadmm_mp_periods(h), exaadmm_mp_periods(h) == 1 ? "" : "s");
    printf("Status  . . . . . . . . . . . . . . . . %s\n", info.status == 1 ? "Solved" : "IterationLimit");
    printf("Objective value  . . . . . . . . . . . %12.6e\n", info.objval);
    printf("Residual  . . . . . . . . . . . . . . . %12.6e\n", info.mismatch);
    printf("Cumulative iterations . . . . . . . . . %5d\n", info.cumul);
    printf("Outer iterations  . . . . . . . . . . . %5d\n", info.outer);
    printf("Time per iteration  . . . . . . . . . . %12.6f (secs/iter)\n", info.cumul > 0 ? info.time_overall / info.cumul : 0.0);
    printf("Overall time  . . . . . . . . . . . . . %12.6f (secs)\n", info.time_overall);
    printf("Generator + branch time (sampled) . . . %12.6f (secs)\n", info.time_x_update);
    printf("Bus time (sampled)  . . . . . . . . . . %12.6f (secs)\n", info.time_xbar_update);
    printf("z, l, residual time (sampled) . . . . . %12.6f (secs)\n", info.time_zl_update);
    /* (the block above prints what the reference prints, 7 significant digits; scripts and tests read the full value here) */
    printf("objval=%.17g mismatch=%.17g\n", info.objval, info.mismatch);
    exaadmm_destroy(h);
    exaadmm_case_free(cs);
    return info.status == 1 ? 0 : 1;
}
