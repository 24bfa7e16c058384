// Host check of fcdiff_amd/csrc/fcd_fastmath.h (K_lik's exp and log) against long-double libm: prints the worst error
// in ulp of each function.  Built and run by tests/test_fastmath.py.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../fcdiff_amd/csrc/fcd_fastmath.h"

static double ulp_err(double got, long double want) {
    if (want == 0) return got == 0 ? 0 : 1e9;
    int e;
    frexpl(want, &e);
    const long double ulp = ldexpl(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 3000000;
    double et[FCD_EXP_CELLS];
    fcd_log_cell lt[FCD_LOG_CELLS];
    fcd_fm_make_tables(et, lt, exp2, log);
    srand48(1);
    double worst_e = 0, worst_l = 0;
    for (long i = 0; i < n; ++i) {
        const double y = (i & 1) ? drand48() * 708.0 : drand48() * drand48() * 40.0;     // results stay normal
        const double u = ulp_err(fcd_exp_neg(y, et), expl(-(long double)y));
        if (u > worst_e) worst_e = u;
    }
    for (long i = 0; i < n; ++i) {
        double x;
        const int mode = (int)(i % 3);
        if (mode == 0) x = ldexp(1.0 + drand48(), (int)(drand48() * 2040) - 1020);      // the whole normal range
        else if (mode == 1) x = 0.9 + 0.2 * drand48();                                   // around 1: no cancellation
        else x = exp(-drand48() * 100);
        if (x < 2.2250738585072014e-308) continue;
        const double u = ulp_err(fcd_log_normal(x, lt), logl((long double)x));
        if (u > worst_l) worst_l = u;
    }
    const int edge_ok = fcd_exp_neg(0.0, et) == 1.0 && fcd_exp_neg(1e4, et) == 0.0 && fcd_exp_neg(1101.0, et) == 0.0 &&
                        fcd_log_normal(1.0, lt) == 0.0 &&
                        fabs(fcd_log_normal(2.2250738585072014e-308, lt) - log(2.2250738585072014e-308)) < 1e-12 &&
                        fabs(fcd_log_normal(1.7976931348623157e308, lt) - log(1.7976931348623157e308)) < 1e-12;
    printf("%.4f %.4f %d\n", worst_e, worst_l, edge_ok);
    return 0;
}
