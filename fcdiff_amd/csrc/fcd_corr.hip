// K_corr: per-subject region x time sample correlation -> edge-major correlations, optional Fisher z.
//
// Not in the reference (its inputs are already correlations, fcdiff/fit.py:20-23): this is the front-end that
// BASELINE.json's north_star names; oracle = numpy.corrcoef ("parity unpinned").
// Fisher z (atanh) is OFF by default: the model's Normal components live on raw correlations clipped to
// [-1, 1] (fcdiff/model.py:213, 236), so every default of the reference assumes untransformed values.
//
// numpy.corrcoef, restated:  X -= mean(X, axis=1);  c = (X X^T) * (1/(T-1));  s = sqrt(diag c);
//                            c /= s[:, None];  c /= s[None, :];  clip to [-1, 1].
// The Gram matrix is an fp64 MFMA product (v_mfma_f64_16x16x4_f64) over 64 x 64 blocks of the lower triangle with
// LDS-staged, double-buffered 64 x 16 panels.  Output out[c][s], c = n(n-1)/2 + m (n > m): the layout of b / bt.
#include "fcd_common.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// mean and 1-sigma of every (subject, region) row: one wave per row
__global__ __launch_bounds__(256) void corr_moments_kernel(const double *__restrict__ ts, int64_t rows, int T,
                                                           double *__restrict__ mean, double *__restrict__ sdev) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double *x = ts + row * T;
    double s = 0.0;
    for (int k = lane; k < T; k += 64) s += x[k];
    s = fcd_wave_sum(s);
    const double mu = s / (double)T;
    double q = 0.0;
    for (int k = lane; k < T; k += 64) {
        const double d = x[k] - mu;
        q += d * d;
    }
    q = fcd_wave_sum(q);
    if (lane == 0) {
        mean[row] = mu;
        sdev[row] = sqrt(q * (1.0 / (double)(T - 1)));     // sqrt(diag(cov)), cov = X X^T * (1/(T-1))
    }
}

// Gram blocks.  grid = subjects x lower-triangle blocks of 64 x 64 regions (dealt per XCD, see below); block = 4 waves, wave (wr, wc) owns the
// 2 x 2 MFMA tiles (2 wr + {0,1}, 2 wc + {0,1}) of the block.  K = T in steps of 16: the two 64 x 16 panels (rows of the
// block's row regions / column regions, centred on the way in) are staged in LDS -- 128 contiguous bytes per region and
// step, each element loaded once per block instead of once per tile -- double-buffered so that the loads of step k+1 fly
// while the 16 MFMAs per wave of step k run.  Panel rows are padded to 18 doubles: the fragment reads (lane = (row i,
// k-quarter q): P[16 t + i][4 g + q]) then fall on 32 distinct bank pairs.
// Fragments of v_mfma_f64_16x16x4_f64: A[i][k] in lane (i = l & 15, k = l >> 4); B[k][j] in lane (j = l & 15, k = l >> 4);
// D[i][j] comes back as 4 doubles per lane: col = l & 15, row = (l >> 4) + 4 r.
constexpr int CB = 64, CK = 16, CLD = 18;
__global__ __launch_bounds__(256) void corr_gram_kernel(const double *__restrict__ ts, const double *__restrict__ mean,
                                                        const double *__restrict__ sdev, int Nreg, int T, int64_t S,
                                                        int n_blocks, int fisher_z, int64_t C, double *__restrict__ tmp) {
    __shared__ __attribute__((aligned(16))) double pa[2][CB * CLD], pb[2][CB * CLD];
    // Workgroup -> (subject, block): consecutive workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 names the
    // XCD class), so ALL blocks of a subject take the same class: the subject's rows (1.9 MB at cfg3, 3.8 MB at cfg5) are
    // fetched into ONE 4 MB L2 and re-read from there by its other blocks, instead of once per XCD (a speed hint only).
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int t = jj % n_blocks;                        // block index -> (I, J), I >= J, lower-triangular row-major
    const int64_t s = (int64_t)(jj / n_blocks) * 8 + xcd;
    if (s >= S) return;
    int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)I * (I + 1) / 2 > t) --I;
    while ((int64_t)(I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const bool diag = I == J;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    // staging: thread -> (row r of the panel, 16-byte piece p of its 128 bytes), rows r and r + 32: eight lanes cover one
    // whole 128-byte line per load instruction (T even: rows are 16-byte aligned; odd T takes the 8-byte path)
    const int prow = tid >> 3, pp = tid & 7;
    const bool even = (T & 1) == 0;
    int ga[2], gb[2];
    bool va[2], vb[2];
    const double *xa[2], *xb[2];
    double ma[2], mb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ga[h] = I * CB + prow + 32 * h;
        gb[h] = J * CB + prow + 32 * h;
        va[h] = ga[h] < Nreg;
        vb[h] = gb[h] < Nreg && !diag;
        xa[h] = ts + (s * Nreg + (va[h] ? ga[h] : 0)) * T;
        xb[h] = ts + (s * Nreg + (vb[h] ? gb[h] : 0)) * T;
        ma[h] = va[h] ? mean[s * Nreg + ga[h]] : 0.0;
        mb[h] = vb[h] ? mean[s * Nreg + gb[h]] : 0.0;
    }
    double2 ra[2], rb[2];
    auto fetch = [&](int k0) {
        const int k = k0 + 2 * pp;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 a, b;
            if (even) {
                const int kc = k < T ? k : T - 2;              // clamped: no branch around the load
                a = *reinterpret_cast<const double2 *>(xa[h] + kc);
                b = *reinterpret_cast<const double2 *>(xb[h] + kc);
            } else {
                const int k0c = k < T ? k : T - 1, k1c = k + 1 < T ? k + 1 : T - 1;
                a = make_double2(xa[h][k0c], xa[h][k1c]);
                b = make_double2(xb[h][k0c], xb[h][k1c]);
            }
            ra[h].x = (va[h] && k < T) ? a.x - ma[h] : 0.0;
            ra[h].y = (va[h] && k + 1 < T) ? a.y - ma[h] : 0.0;
            rb[h].x = (vb[h] && k < T) ? b.x - mb[h] : 0.0;
            rb[h].y = (vb[h] && k + 1 < T) ? b.y - mb[h] : 0.0;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<double2 *>(&pa[buf][(prow + 32 * h) * CLD + 2 * pp]) = ra[h];
            if (!diag) *reinterpret_cast<double2 *>(&pb[buf][(prow + 32 * h) * CLD + 2 * pp]) = rb[h];
        }
    };
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
    // tiles of this wave that hold anything: rows below Nreg, and (diagonal block) not strictly above the diagonal
    const int i16 = l & 15, kq = l >> 4;
    bool on[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int tr = 2 * wr + a, tc = 2 * wc + b;
            on[a][b] = (I * CB + tr * 16 < Nreg) && (J * CB + tc * 16 < Nreg) && (!diag || tc <= tr);
        }
    const bool any_on = on[0][0] || on[0][1] || on[1][0] || on[1][1];
    fetch(0);
    put(0);
    __syncthreads();
    const int steps = (T + CK - 1) / CK;
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        if (st + 1 < steps) fetch((st + 1) * CK);
        if (any_on) {
            const double *A = pa[cur];
            const double *B = diag ? pa[cur] : pb[cur];
#pragma unroll
            for (int g = 0; g < CK / 4; ++g) {
                double fa[2], fb[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = A[((2 * wr + a) * 16 + i16) * CLD + g * 4 + kq];
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[b] = B[((2 * wc + b) * 16 + i16) * CLD + g * 4 + kq];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (on[a][b]) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        if (st + 1 < steps) put(cur ^ 1);
        __syncthreads();
    }
    const double inv = 1.0 / (double)(T - 1);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (!on[a][b]) continue;
            const int m = J * CB + (2 * wc + b) * 16 + i16;                  // column = the smaller region index of the pair
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = I * CB + (2 * wr + a) * 16 + kq + 4 * r;       // row
                if (n < Nreg && m < n) {
                    double c = acc[a][b][r] * inv;
                    c /= sdev[s * Nreg + n];
                    c /= sdev[s * Nreg + m];
                    c = (c != c) ? c : fmin(fmax(c, -1.0), 1.0);      // (a constant series: 0 / 0 = NaN like numpy.corrcoef; fmin / fmax would drop it)
                    if (fisher_z) c = atanh(c);
                    tmp[s * C + (fcd_tri(n) + m)] = c;                       // 16 lanes = 16 consecutive edges = 128 bytes
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// One workgroup per (subject, slice of the time axis): the whole lower triangle of the subject's Gram matrix.
//
// The blocked kernel above reads every row of a subject 4 times over (64 x 64 blocks: 800 row loads for 200 regions) and
// needs the means before it starts (a pass of its own over the input).  Here a workgroup of 16 waves stages ALL rows of
// its subject, 16 time steps at a time, and owns all 16 x 16 tiles of the lower triangle (TPW per wave, consecutive in
// row-major order): every input byte is loaded ONCE.  Centring is by the row's first sample x0 instead of its mean --
//     sum (x - mx)(y - my) = sum (x - x0)(y - y0) - (sum (x - x0)) (sum (y - y0)) / T
// -- a shift that removes the cancellation the raw second moment would have (the shifted values are of the size of the
// deviations), so one pass gives Gram matrix and row sums; the deviations are the roots of the Gram diagonal, as in
// numpy.corrcoef.  The time axis is cut in KS slices (S subjects alone would leave most CUs idle): each workgroup
// writes its partial tiles, the last of a subject to finish (a ticket) adds the slices IN SLICE ORDER (same bits
// whoever comes last), normalises and writes the subject-major row of tmp.
// ---------------------------------------------------------------------------------------------
template <int TPW>
__global__ __launch_bounds__(1024) void corr_gram_subject_kernel(const double *__restrict__ ts, int Nreg, int T, int64_t S, int PMAX,
                                                                 int64_t C, int fisher_z, double *__restrict__ part,
                                                                 unsigned *__restrict__ ticket, double *__restrict__ tmp) {
    extern __shared__ __attribute__((aligned(16))) double csm[];
    const int RT = (Nreg + 15) >> 4, RP = RT * 16, NT = RT * (RT + 1) / 2;
    // steps of SK = 32 samples (two halves of 16, staged one after the other): half as many barriers per MFMA as with 16
    constexpr int SK = 32, SLD = 34;        // (rows padded to 34 doubles: the fragment reads stay conflict-free)
    double *dg = csm + 2 * RP * SLD, *sums = dg + RP, *sd = sums + RP;     // (csm: panels [2][RP][SLD] first)
    __shared__ int sh_last;
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = l & 15, kq = l >> 4;
    // the wave's tiles (tr >= tc), wave-uniform
    int tr[TPW], tc[TPW];
    bool on[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = w * TPW + j;
        on[j] = t < NT;
        const int tt = on[j] ? t : 0;
        int r = (int)((sqrt(8.0 * (double)tt + 1.0) - 1.0) * 0.5);
        while (r * (r + 1) / 2 > tt) --r;
        while ((r + 1) * (r + 2) / 2 <= tt) ++r;
        tr[j] = __builtin_amdgcn_readfirstlane(r);
        tc[j] = __builtin_amdgcn_readfirstlane(tt - r * (r + 1) / 2);
    }
    // staging: piece idx = tid + 1024 i -> (row = idx / 8, 16-byte piece pp = idx % 8 of the row's 128 bytes per step)
    constexpr int NPI = 2;                      // RP * 8 <= 2048 pieces (host: RP <= 256)
    const int pp = tid & 7;
    const bool even = (T & 1) == 0;
    // workgroup -> (subject, slice ks of KS of the time axis)
    const int KS = PMAX;
    const int s = (int)blockIdx.x / KS, ks = (int)blockIdx.x % KS;
    const int steps_all = (T + SK - 1) / SK, per = (steps_all + KS - 1) / KS;
    const int st_lo = ks * per, st_hi = (st_lo + per < steps_all) ? st_lo + per : steps_all;
    const double *xp[NPI];
    double x0[NPI], rs[NPI];
    bool va[NPI];
    int prow[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        prow[i] = (tid >> 3) + 128 * i;
        va[i] = prow[i] < Nreg;
        xp[i] = ts + ((int64_t)s * Nreg + (va[i] ? prow[i] : 0)) * T;
        x0[i] = va[i] ? xp[i][0] : 0.0;
        rs[i] = 0.0;
    }
    // fetch only ASKS for the step's values; they are shifted, masked and summed in put, behind the MFMAs of the step
    // before (anything done to them in fetch would wait for the loads in front of those MFMAs)
    double2 raw[NPI];
    int kf = 0;
    auto fetch = [&](int st, int h) {              // half h (16 samples) of step st
        kf = st * SK + h * 16 + 2 * pp;
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            if (even) {
                const int kc = kf < T ? kf : T - 2;            // clamped: no branch around the load
                raw[i] = *reinterpret_cast<const double2 *>(xp[i] + kc);
            } else {
                const int k0c = kf < T ? kf : T - 1, k1c = kf + 1 < T ? kf + 1 : T - 1;
                raw[i] = make_double2(xp[i][k0c], xp[i][k1c]);
            }
        }
    };
    auto put = [&](int buf, int h) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            double2 v;
            v.x = (va[i] && kf < T) ? raw[i].x - x0[i] : 0.0;
            v.y = (va[i] && kf + 1 < T) ? raw[i].y - x0[i] : 0.0;
            rs[i] += v.x + v.y;
            if (prow[i] < RP) *reinterpret_cast<double2 *>(&csm[buf * RP * SLD + prow[i] * SLD + h * 16 + 2 * pp]) = v;
        }
    };
    double4_t acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
    // Two panels in LDS.  Step st (32 samples): one barrier (panel st % 2 is complete, nobody reads the other one any
    // more), the fragments of the first eighth, then the 48 MFMAs of the step -- behind every MFMA the fragments of the
    // SAME tile for the next eighth (asked for a whole turn of the wave's six tiles before they are used) -- and, spread
    // over the eighths, the staging of step st + 1: its first 16 samples are asked for behind eighth 0 and shifted, summed
    // and parked in the other panel behind eighth 3, the second 16 behind eighths 4 and 7 (the same four registers
    // twice).  All of that issues while the matrix pipe works through the MFMAs of the SIMD's four waves; only the first
    // fragments of a step wait for the LDS.
    double fa[TPW], fb[TPW];
    auto frags = [&](int buf, int j, int g) {
        // (a slot beyond the last tile runs tile 0 again and is dropped at the end: no branch around an MFMA)
        const double *P = csm + buf * RP * SLD + (i16 * SLD + kq);
        fa[j] = P[tr[j] * (16 * SLD) + g * 4];
        fb[j] = P[tc[j] * (16 * SLD) + g * 4];
    };
    if (st_lo < st_hi) {
        fetch(st_lo, 0);
        put(0, 0);
        fetch(st_lo, 1);
        put(0, 1);
    }
    for (int st = st_lo; st < st_hi; ++st) {
        const int cur = (st - st_lo) & 1;
        const bool more = st + 1 < st_hi;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TPW; ++j) frags(cur, j, 0);
#pragma unroll
        for (int g = 0; g < SK / 4; ++g) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[j], fb[j], acc[j], 0, 0, 0);
                if (g + 1 < SK / 4) frags(cur, j, g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
                if (g == 0) fetch(st + 1, 0);
                if (g == 3) put(cur ^ 1, 0);
                if (g == 4) fetch(st + 1, 1);
                if (g == 7) put(cur ^ 1, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    // row sums of the slice: the 8 pieces of a row sit in 8 consecutive lanes
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        double v = rs[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if (pp == 0 && prow[i] < RP) sums[prow[i]] = v;
    }
    __syncthreads();
    const int64_t psz = (int64_t)NT * 256 + RP;
    if (KS > 1) {
        // The slices meet in memory.  Every partial value is written and read with agent-scope accesses (straight to / from
        // the memory side: the L2s of the XCDs are not coherent with one another), the ticket is drawn once they are
        // acknowledged -- no fence: a device-scope fence writes back / invalidates a whole L2, and one per wave of every
        // workgroup cost more than the Gram product itself (215 of 350 us).
        double *mine = part + ((int64_t)s * PMAX + ks) * psz;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            if (on[j]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(mine + (int64_t)(w * TPW + j) * 256 + r * 64 + l, acc[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        if (tid < RP) __hip_atomic_store(mine + (int64_t)NT * 256 + tid, sums[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every wave waits for the acknowledgement of ITS stores before the barrier: a barrier on this target waits for
        // LDS traffic only (lgkmcnt), so without this the ticket could reach the memory side before another wave's
        // partial tiles (the in-order role of the r pass does the same before its mark)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&ticket[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)KS - 1u;
            if (last) __hip_atomic_store(&ticket[s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_last = last;
        }
        __syncthreads();
        if (!sh_last) return;
        // the slices in slice order
        const double *p0 = part + (int64_t)s * PMAX * psz;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            if (!on[j]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *q0 = p0 + (int64_t)(w * TPW + j) * 256 + r * 64 + l;
                double a = __hip_atomic_load(q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int q = 1; q < KS; ++q) a += __hip_atomic_load(q0 + q * psz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc[j][r] = a;
            }
        }
        if (tid < RP) {
            double v = __hip_atomic_load(p0 + (int64_t)NT * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 1; q < KS; ++q) v += __hip_atomic_load(p0 + q * psz + (int64_t)NT * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sums[tid] = v;
        }
    }
    // diagonal of the Gram matrix -> deviations
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (on[j] && tr[j] == tc[j]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kq + 4 * r == i16) dg[tr[j] * 16 + i16] = acc[j][r];
        }
    }
    __syncthreads();
    const double inv = 1.0 / (double)(T - 1), invT = 1.0 / (double)T;
    // 1 / sqrt(diag(cov)): numpy divides by the two deviations in turn; two multiplications by their reciprocals differ from
    // that by an ulp or two (the tests allow 1e-11) and spare every lane 48 fp64 divisions at the tail of the kernel
    if (tid < RP) sd[tid] = 1.0 / sqrt((dg[tid] - sums[tid] * sums[tid] * invT) * inv);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (!on[j]) continue;
        const int m = tc[j] * 16 + i16;                                  // column = the smaller region index of the pair
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tr[j] * 16 + kq + 4 * r;                       // row
            if (n < Nreg && m < n) {
                double c = (acc[j][r] - sums[n] * sums[m] * invT) * inv;
                c *= sd[n];
                c *= sd[m];
                c = (c != c) ? c : fmin(fmax(c, -1.0), 1.0);             // (a constant series: 0 x inf = NaN, like numpy.corrcoef's 0 / 0)
                if (fisher_z) c = atanh(c);
                tmp[(int64_t)s * C + (fcd_tri(n) + m)] = c;              // 16 lanes = 16 consecutive edges = 128 bytes
            }
        }
    }
}

// tmp (S, C) -> out (C, S): the layout of b / bt (edge-major, subjects fastest).  The Gram kernel writes subject-major
// rows (128 contiguous bytes per 16 lanes); written straight into (C, S) every value would be an 8-byte store 8 S bytes
// from the next one -- 40 M partial-line writes at cfg5.  32 x 32 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void corr_transpose_kernel(const double *__restrict__ tmp, int64_t S, int64_t C,
                                                             double *__restrict__ out) {
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, s0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8 threads
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t s = s0 + ty + j, c = c0 + tx;
        if (s < S && c < C) tile[ty + j][tx] = tmp[s * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t c = c0 + ty + j, s = s0 + tx;
        if (s < S && c < C) out[c * S + s] = tile[tx][ty + j];
    }
}

}  // namespace

extern "C" int fcd_corr_edges(fcd_ctx *ctx, const double *ts, int64_t S, int64_t Nreg, int64_t T, int fisher_z, double *out,
                              fcd_stream stream) {
    if (!ctx || !ts || !out) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_corr_edges: null pointer");
    if (S < 1 || Nreg < 2 || T < 2) return fcd_fail(ctx, FCD_ERR_SHAPE, "need S >= 1, Nreg >= 2, T >= 2 (Nreg=%lld, T=%lld)", Nreg, T);
    if (S > 65535 || Nreg > 46340 || T > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: shape too large");
    const int64_t rows = S * Nreg, C = fcd_tri(Nreg);
    hipStream_t s = (hipStream_t)stream;
    if ((S + 31) / 32 > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: grid too large");
    // Up to 208 regions (16 waves x 6 tiles of 16 x 16 cover the lower triangle): one workgroup per (subject, time slice)
    constexpr int TPW = 6;
    const int RT = (int)((Nreg + 15) / 16), RP = RT * 16, NT = RT * (RT + 1) / 2;
    if (NT <= 16 * TPW && ctx->knobs.corr_form != 1) {
        // slices of the time axis: about one workgroup per CU (S subjects alone would leave most CUs idle), no slice
        // shorter than 2 steps of 32 samples
        const int64_t steps_all = (T + 31) / 32;                        // (the kernel's steps: 32 samples)
        int64_t KS = ctx->num_cu / S;
        if (KS > steps_all / 2) KS = steps_all / 2;
        if (KS < 1) KS = 1;
        if (KS > 16) KS = 16;
        const int64_t NWG = S * KS;
        const int PMAX = (int)KS;
        const size_t psz = (size_t)NT * 256 + RP;
        const size_t part_bytes = (size_t)S * PMAX * psz * sizeof(double);
        int rc = fcd_ws_reserve(ctx, part_bytes + (size_t)S * C * sizeof(double));
        if (rc) return rc;
        // the tickets live in the context, zero between launches (the last taker of a subject puts its ticket back):
        // no memset launch per call.  Growing them synchronises, like growing the scratch.
        if (ctx->corr_tickets_n < (size_t)S) {
            FCD_HIP_TRY(hipDeviceSynchronize());
            if (ctx->corr_tickets) (void)hipFree(ctx->corr_tickets);
            ctx->corr_tickets = nullptr;
            ctx->corr_tickets_n = 0;
            FCD_HIP_TRY(hipMalloc(&ctx->corr_tickets, (size_t)S * sizeof(unsigned)));
            FCD_HIP_TRY(hipMemset(ctx->corr_tickets, 0, (size_t)S * sizeof(unsigned)));
            ctx->corr_tickets_n = (size_t)S;
        }
        unsigned *ticket = (unsigned *)ctx->corr_tickets;
        double *part = (double *)ctx->ws;
        double *tmp = (double *)((char *)ctx->ws + part_bytes);
        const size_t shmem = ((size_t)2 * RP * 34 + 3 * RP) * sizeof(double);      // (panels [2][RP][34]: the kernel's SLD)
        rc = fcd_lds_attr(ctx, FCD_KA_CORR, reinterpret_cast<const void *>(&corr_gram_subject_kernel<TPW>), shmem);
        if (rc) return rc;
        hipLaunchKernelGGL(corr_gram_subject_kernel<TPW>, dim3((unsigned)NWG), dim3(1024), shmem, s, ts, (int)Nreg, (int)T, S,
                           PMAX, C, fisher_z ? 1 : 0, part, ticket, tmp);
        FCD_LAUNCH_CHECK();
        hipLaunchKernelGGL(corr_transpose_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((S + 31) / 32)), dim3(256), 0, s, tmp, S, C, out);
        FCD_LAUNCH_CHECK();
        return FCD_OK;
    }
    // more regions: 64 x 64 blocks, the means and deviations in a pass of their own
    // workspace: means and deviations of the rows, then the subject-major copy of the result (grows the context's
    // scratch, which synchronises, the first time a shape needs it)
    int rc = fcd_ws_reserve(ctx, ((size_t)rows * 2 + (size_t)S * C) * sizeof(double));
    if (rc) return rc;
    double *mean = (double *)ctx->ws, *sdev = mean + rows, *tmp = sdev + rows;
    hipLaunchKernelGGL(corr_moments_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ts, rows, (int)T, mean, sdev);
    FCD_LAUNCH_CHECK();
    const int64_t nb = (Nreg + CB - 1) / CB;
    const int64_t blocks = nb * (nb + 1) / 2;
    const int64_t grid = (S + 7) / 8 * 8 * blocks;
    if (grid > INT32_MAX) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: S * blocks too large");
    hipLaunchKernelGGL(corr_gram_kernel, dim3((unsigned)grid), dim3(256), 0, s, ts, mean, sdev, (int)Nreg, (int)T, S, (int)blocks,
                       fisher_z ? 1 : 0, C, tmp);
    FCD_LAUNCH_CHECK();
    if ((C + 31) / 32 > INT32_MAX || (S + 31) / 32 > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_corr_edges: grid too large");
    hipLaunchKernelGGL(corr_transpose_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((S + 31) / 32)), dim3(256), 0, s, tmp, S, C, out);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
