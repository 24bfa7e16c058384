// Table-driven exp(-y) and log for K_lik (fcd_lik.hip).  Host + device: tests/test_fastmath.py compiles this header
// with g++ and checks both functions against libm over their whole input ranges (<= 2 ulp).
//
// The kernel is ALU-bound on its 3 exponentials and 9 logarithms per (edge, patient) item; ocml's exp / log are ~35 and
// ~70 fp64 instructions.  These are ~18 and ~22: they can afford that because K_lik's arguments are special --
// exp only ever sees -z*z/2 <= 0, and the logs see positive, finite, NORMAL numbers (the caller checks that once
// per item and sends the rare item near the underflow edge down the generic path).
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef FCD_FM_HD
#ifdef __HIPCC__
#define FCD_FM_HD __host__ __device__ inline
#else
#define FCD_FM_HD inline
#endif
#endif

#define FCD_EXP_CELLS 64
#define FCD_LOG_CELLS 512

struct fcd_log_cell {
    double inv, lg;    // 1/m_i (rounded), -log(inv)
};

FCD_FM_HD uint64_t fcd_fm_bits(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
FCD_FM_HD double fcd_fm_double(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}

// ---- tables (host side; the context copies them to the device once) ----
// exp: T[j] = 2^(-j/64), j = 0..63
// log: cell i = top 9 mantissa bits of the [1,2)-normalised argument, m_i = cell midpoint; cells 256..511 (m >= 1.5) are
//      halved into [0.75, 1); the two cells around 1 (i = 0 and i = 511) use m_i = 1, so there is no cancellation near 1.
//      lg is the log of the ROUNDED reciprocal's inverse, so  log x = e ln2 + lg + log1p(m * inv - 1)  stays an identity.
static inline void fcd_fm_make_tables(double *exp_tab, fcd_log_cell *log_tab, double (*exp2_fn)(double), double (*log_fn)(double)) {
    for (int j = 0; j < FCD_EXP_CELLS; ++j) exp_tab[j] = exp2_fn(-(double)j / FCD_EXP_CELLS);
    for (int i = 0; i < FCD_LOG_CELLS; ++i) {
        double m = 1.0 + (i + 0.5) / FCD_LOG_CELLS;
        if (i >= FCD_LOG_CELLS / 2) m *= 0.5;
        if (i == 0 || i == FCD_LOG_CELLS - 1) m = 1.0;
        log_tab[i].inv = 1.0 / m;
        log_tab[i].lg = -log_fn(log_tab[i].inv);
    }
}

// exp(-y), y >= 0 (y = z*z/2).  -y = -(k/64) ln2 + r, |r| <= ln2/128; exp(r) by its Taylor polynomial to r^6
// (r^7/5040 < 2^-63); 2^(-k/64) = T[k & 63] * 2^-(k >> 6).  <= 1.5 ulp while the result is normal; below that the
// caller recomputes with the reference's own operations.  y > 1100 returns 0 (true value < 2^-1586).
#ifdef __HIPCC__
#define FCD_FM_FMA(a, b, c) __builtin_fma(a, b, c)
#define FCD_FM_RINT(x) __builtin_rint(x)
#define FCD_FM_LDEXP(x, e) __builtin_ldexp(x, e)
#else
#include <math.h>
#define FCD_FM_FMA(a, b, c) fma(a, b, c)
#define FCD_FM_RINT(x) rint(x)
#define FCD_FM_LDEXP(x, e) ldexp(x, e)
#endif

template <typename TAB>
FCD_FM_HD double fcd_exp_neg(double y, TAB exp_tab) {
    if (!(y <= 1100.0)) return (y != y) ? y : 0.0;
    const double kf = FCD_FM_RINT(y * 92.33248261689365676830);          // 64 / ln 2
    const int k = (int)kf;
    // r = kf * (ln2/64) - y, with ln2/64 split so that kf * hi is exact (hi keeps 34 bits, kf < 2^17)
    double r = FCD_FM_FMA(kf, 1.083042469599604373798e-02, -y);       // ln2/64 hi
    r = FCD_FM_FMA(kf, 2.531017216665087694876e-13, r);               // ln2/64 lo  (hi + lo = ln2/64 to 2e-29)
    double p = 1.0 / 720.0;
    p = FCD_FM_FMA(p, r, 1.0 / 120.0);
    p = FCD_FM_FMA(p, r, 1.0 / 24.0);
    p = FCD_FM_FMA(p, r, 1.0 / 6.0);
    p = FCD_FM_FMA(p, r, 0.5);
    p = FCD_FM_FMA(p, r, 1.0);
    const double t = exp_tab[k & (FCD_EXP_CELLS - 1)];
    const double v = FCD_FM_FMA(t * r, p, t);                          // t * (1 + r p)
    return FCD_FM_LDEXP(v, -(k >> 6));
}

// log x for x POSITIVE, FINITE and NORMAL (the caller guarantees it).  |r| <= 2^-9, log1p(r) to r^7 (r^8/8 < 2^-75 r).
template <typename TAB>
FCD_FM_HD double fcd_log_normal(double x, TAB log_tab) {
    const uint64_t bits = fcd_fm_bits(x);
    const int idx = (int)((bits >> 43) & (FCD_LOG_CELLS - 1));      // top 9 mantissa bits
    int e = (int)(bits >> 52) - 1023;
    // mantissa in [1, 2); cells >= 256 (m >= 1.5) are halved into [0.75, 1) and the exponent bumped
    const uint64_t hi_half = (uint64_t)(idx >> 8);                  // 0 or 1
    const uint64_t mb = ((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull) - (hi_half << 52);
    e += (int)hi_half;
    const double m = fcd_fm_double(mb);
    const double inv = log_tab[idx].inv, lg = log_tab[idx].lg;
    const double r = FCD_FM_FMA(m, inv, -1.0);
    // log1p(r) = r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6 + r^7/7
    double p = 1.0 / 7.0;
    p = FCD_FM_FMA(p, r, -1.0 / 6.0);
    p = FCD_FM_FMA(p, r, 0.2);
    p = FCD_FM_FMA(p, r, -0.25);
    p = FCD_FM_FMA(p, r, 1.0 / 3.0);
    p = FCD_FM_FMA(p, r, -0.5);
    const double r2 = r * r;
    const double de = (double)e;
    const double hi = FCD_FM_FMA(de, 6.93147180369123816490e-01, lg);        // e*ln2_hi + log m_i (ln2_hi has 32 trailing zero bits)
    const double lo = FCD_FM_FMA(de, 1.90821492927058770002e-10, FCD_FM_FMA(p, r2, r));   // e*ln2_lo + log1p(r)
    return hi + lo;
}
