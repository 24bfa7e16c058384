// r step of the collapsed Gibbs sampler: redraw every r_nu given f, regions in order 0..Nreg-1.
// Conditional = fcdiff/fit.py:187-194 at one-hot q_F, q_R:
//   s0 = ln(1-pi) + sum_{m != n} lM[c, u, f_c, r_mu ? 2 : 0],  s1 = ln pi + sum_{m != n} lM[c, u, f_c, r_mu ? 1 : 2]
// and only d = s1 - s0 enters the draw:  r_nu = 1  <=>  logit(x) < d.
//
// The scan over n is a Gauss-Seidel sweep: r_n sees the NEW r_m for m < n and the OLD r_m for m > n.
// It is organised like a blocked forward substitution with a one-block look-ahead.  Regions are cut into blocks
// of R_NB = 16; block step s does two independent things, in ONE launch (gibbs_r_step_kernel):
//   * role P(s): for every n of block s, the sum over all m outside blocks s-1 and s -- new r below block s-1, old r
//     above block s, all final before the launch starts; fully parallel over (n, patient, chain); the region-major
//     rows lMd[u][n][:] are staged in LDS and shared by all chain words of the workgroup, so the table is read once
//     per pass.  The same workgroups make the draw thresholds of their (region, patient pair) and store
//     e = ln(pi/(1-pi)) + sum - logit(x);
//   * role D(s-1): for each (patient, chain word) walk the 16 regions of block s-1 in order, adding to e the terms
//     against block s-2 (redrawn by the previous launch) and against the own block from LDS copies of the two
//     tiles, and draw r_n = [v > 0]; one wave per (patient, chain word), nothing but the 1-bit decision links one
//     region to the next.
// ceil(Nreg / 16) + 1 launches per pass; the serial part rides beside the parallel part of the next block.
// gibbs_r_pipe_kernel is the same pass in ONE launch (every workgroup resident, hand-over through marks and sentinels in
// device memory): the default wherever its grid fits the device at once.
//
// lMd (U, Nreg, Nreg, 3, 2) is a region-major DIFFERENCE table made once per table build:
//   lMd[u][n][m][k][t] = t ? lM[c,u,k,1] - lM[c,u,k,2] : lM[c,u,k,2] - lM[c,u,k,0],   c = edge(n, m)
// i.e. the contribution of region m to d for f_c = k and r_mu = t; edge() is the SAME ordered-pair edge id
// the reference uses (fit.py:186 calls nm_to_c(n, m) for every ordered pair in 'reference' mode), so that
// quirk is baked into the table.  One term (a PAIR of regions, see the pair records below) = one 8-byte LDS read
// + one fp64 add + two integer instructions.
#include <stdlib.h>

#include "fcd_common.h"

namespace {

constexpr int R_NB = 16;   // regions per diagonal block (even: both halves of a Philox block stay inside)

// ---------------------------------------------------------------------------------------------
// region-major difference table: one thread per (u, n, m) record of 6 doubles
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void region_tables_kernel(const double *__restrict__ lM, int Nreg, int U, int mode,
                                                            double *__restrict__ lMd) {
    const int64_t total = (int64_t)U * Nreg * Nreg;
    for (int64_t rec = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; rec < total; rec += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(rec % Nreg);
        const int n = (int)((rec / Nreg) % Nreg);
        const int u = (int)(rec / ((int64_t)Nreg * Nreg));
        double *o = lMd + rec * 6;
        if (m == n) {
#pragma unroll
            for (int x = 0; x < 6; ++x) o[x] = 0.0;
        } else {
            const double *p = lM + (fcd_pair_to_edge(n, m, mode) * U + u) * 9;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[k * 2 + 0] = p[k * 3 + 2] - p[k * 3 + 0];   // r_m = 0: lM[k,2] - lM[k,0]   fit.py:188-194
                o[k * 2 + 1] = p[k * 3 + 1] - p[k * 3 + 2];   // r_m = 1: lM[k,1] - lM[k,2]
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-pass packing of the chain state into the forms the blocked kernels read with one coalesced load per 16
// regions (= 8 PAIRS of regions (2p, 2p+1)).  A pair record in LDS is [q = 3 f(n,m) + f(n,m+1)][tt = r_m + 2 r_m+1]
// doubles, so the byte offset of a term is (q << 5) | (tt << 3) = ((q << 2) | tt) << 3: the words carry ONE BYTE per
// pair, the f part and the r part of that byte are OR-ed once per block, and a term costs two integer instructions
// (extract the byte, shift-add the tile base).
//   f_S[w][n][b][lane]   uint2: byte p (x: pairs 0-3, y: pairs 4-7) = q << 2 of block b
//   r_S[w][u][b][lane]   uint2: byte p = tt of block b, before the pass; r_Sn likewise, the blocks redrawn in the pass
//   (the in-order part D reads the same bytes: the old ones of its own block from r_S, the redrawn ones of the block before from r_Sn)
// (f of a region beyond Nreg or of m == n counts as 0; those table records are zero.)
// ---------------------------------------------------------------------------------------------
// 8 two-bit fields of x -> the low two bits of 8 bytes (x: fields 0-3, y: fields 4-7)
__device__ __forceinline__ uint2 spread2(uint32_t x16) {
    uint32_t lo = x16 & 0xFFu, hi = (x16 >> 8) & 0xFFu;
    lo = (lo | (lo << 12)) & 0x000F000Fu;
    hi = (hi | (hi << 12)) & 0x000F000Fu;
    lo = (lo | (lo << 6)) & 0x03030303u;
    hi = (hi | (hi << 6)) & 0x03030303u;
    return make_uint2(lo, hi);
}
// LDS address of the dynamic array: 0.  The kernels of this file declare no static LDS (r_static_lds_check asks the
// runtime before the first launch), so the array starts the workgroup's allocation -- and a term's address is the byte
// offset inside the record plus an immediate, with no add of a base the compiler cannot see through.
__device__ __forceinline__ uint32_t lds_base0(const double *) { return 0u; }
// byte p of the pair word -> byte offset inside the pair record ((q << 2 | tt) << 3)
__device__ __forceinline__ uint32_t pair_off(uint2 z, int p) {
    const uint32_t w = p < 4 ? z.x : z.y;
    const int s = 8 * (p & 3);
    return s == 0 ? (w << 3) & 0x7F8u : (w >> (s - 3)) & 0x7F8u;
}

// the same from 6 bits of the byte (q << 2 | tt <= 35): whatever sits in bits 6-7 is ignored
__device__ __forceinline__ uint32_t pair_off6(uint2 z, int p) {
    const uint32_t w = p < 4 ? z.x : z.y;
    const int s = 8 * (p & 3);
    return s == 0 ? (w << 3) & 0x1F8u : (w >> (s - 3)) & 0x1F8u;
}

// (w, n, b) wave-uniform: all index arithmetic is scalar, 32-bit (C * 64 fits an int, checked by the host)
// SQ: f_state is the square copy [w][n][m][lane] (rows contiguous in m) instead of the edge-major state
template <bool SQ>
__device__ __forceinline__ void pack_f_item(const uint8_t *__restrict__ f_state, int Nreg, int NBLK, int C32, int mode,
                                            uint2 *__restrict__ f_S, int w, int n, int b, int lane,
                                            uint8_t *__restrict__ lds_wave) {
    const uint8_t *__restrict__ fw = SQ ? f_state + ((int64_t)w * Nreg + n) * Nreg * 64 : f_state + (int64_t)w * C32 * 64;
    const int tn = (n * (n - 1)) >> 1;
    const uint32_t sh = 8u * (uint32_t)(lane & 3);
    // all loads first (from a clamped, always valid edge: a guard around a load is a branch and a wait per load),
    // the regions that do not exist or are n itself are masked afterwards
    constexpr int NB2 = SQ ? 2 : 1;          // blocks per wave: b, b + 1 (square copy: both loads in flight together)
    uint32_t k[NB2][R_NB];
    if (SQ) {
        // The 16 regions of a block are ONE contiguous kilobyte of the square copy: a single 16-byte load per lane
        // fetches it (lane L: region L/4, chains 16 (L%4) ..), the wave's LDS slice turns it round (region j, chain
        // lane = byte j*64 + lane).  One memory instruction instead of 16.
        uint4 q[NB2];
#pragma unroll
        for (int x = 0; x < NB2; ++x) {
            const int ml = (b + x) * R_NB + (lane >> 2);
            q[x] = *reinterpret_cast<const uint4 *>(fw + (uint32_t)((ml < Nreg ? ml : 0) * 64 + (lane & 3) * 16));
        }
#pragma unroll
        for (int x = 0; x < NB2; ++x) *reinterpret_cast<uint4 *>(lds_wave + x * (R_NB * 64) + lane * 16) = q[x];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int x = 0; x < NB2; ++x)
#pragma unroll
            for (int j = 0; j < R_NB; ++j) k[x][j] = lds_wave[x * (R_NB * 64) + j * 64 + lane];
    } else {
#pragma unroll
        for (int j = 0; j < R_NB; ++j) {
            const int m = b * R_NB + j;
            const bool on = m < Nreg && m != n;
            const int mm = on ? m : (n > 0 ? 0 : 1);
            const int e = (mode == FCD_EDGE_REFERENCE || n > mm) ? tn + mm : ((mm * (mm - 1)) >> 1) + n;   // fcd_pair_to_edge
            // (dword loads, each shared by 4 lanes, then the lane's byte: one-byte-per-lane loads run several times slower)
            k[0][j] = (*reinterpret_cast<const uint32_t *>(fw + (uint32_t)(e * 64 + (lane & ~3))) >> sh) & 0xffu;
        }
    }
#pragma unroll
    for (int x = 0; x < NB2; ++x) {
        if (b + x >= NBLK) break;
        uint32_t v[2] = {0u, 0u};
#pragma unroll
        for (int p = 0; p < R_NB / 2; ++p) {
            const int m0 = (b + x) * R_NB + 2 * p, m1 = m0 + 1;
            const uint32_t k0 = (m0 < Nreg && m0 != n) ? k[x][2 * p] : 0u;
            const uint32_t k1 = (m1 < Nreg && m1 != n) ? k[x][2 * p + 1] : 0u;
            v[p >> 2] |= ((k0 * 3u + k1) << 2) << (8 * (p & 3));
        }
        f_S[(((int64_t)w * Nreg + n) * NBLK + b + x) * 64 + lane] = make_uint2(v[0], v[1]);
    }
}

// r words of (w, u, b): bit j = r_{16 b + j, u}, and the same bits one byte per pair of regions
// (pipe: the pipelined one-launch pass follows -- clear the block's "final" mark and put the "no value yet" NaN
// into the two panel buffers, 16 rows each, blocks 0 and 1 doing one buffer each)
struct r_pipe_init {
    uint32_t *marks;
    double *P[2];
};
constexpr unsigned long long R_SENT = 0x7FF8DEADBEEF0001ull;      // "no panel value here yet" (a NaN no sum can produce)
__device__ __forceinline__ void pack_r_item(const uint64_t *__restrict__ r_bits, int Nreg, int U, int NBLK,
                                            uint2 *__restrict__ r_S, int w, int u, int b, int lane,
                                            const r_pipe_init &pipe) {
    if (pipe.marks) {
        if (lane == 0) pipe.marks[((int64_t)w * U + u) * NBLK + b] = 0u;
        if (b < 2 && pipe.P[b]) {                        // (a single block never uses the second buffer)
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(pipe.P[b]) + (((int64_t)w * U + u) * R_NB) * 64 + lane;
#pragma unroll
            for (int i = 0; i < R_NB; ++i) dst[i * 64] = R_SENT;
        }
    }
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < R_NB; ++j) {
        const int m = b * R_NB + j;
        const uint64_t word = r_bits[((int64_t)w * Nreg + (m < Nreg ? m : Nreg - 1)) * U + u];    // clamped: no branch per load
        v |= (m < Nreg ? (uint32_t)((word >> lane) & 1ull) : 0u) << j;
    }
    const int64_t o = (((int64_t)w * U + u) * NBLK + b) * 64 + lane;
    r_S[o] = spread2(v);
}

// grid (ceil(NBLK / (4 FB)), Nreg + U, GW): one wave per (w, n, FB blocks from b) resp. (w, u, FB blocks), no index
// division; FB = 2 with the square copy, else 1.
// y < Nreg: f words of region n = y;  y >= Nreg: r words of patient u = y - Nreg (r_bits == nullptr: f words only).
template <bool SQ>
__global__ __launch_bounds__(256) void pack_f_kernel(const uint8_t *__restrict__ f_state, int Nreg, int NBLK, int C32,
                                                     int mode, uint2 *__restrict__ f_S, const uint64_t *__restrict__ r_bits,
                                                     int U, uint2 *__restrict__ r_S,
                                                     const r_pipe_init pipe, const fcd_tally_f tf) {
    constexpr int FB = SQ ? 2 : 1;
    __shared__ __attribute__((aligned(16))) uint8_t turn[4][FB * R_NB * 64];      // one kilobyte per wave and block (square-copy form)
    if ((int)blockIdx.y >= Nreg + U) {
        // rows beyond the last patient: the f half of the sweep's tally (edge-major f state, nothing to do with the packing)
        const int lin = (((int)blockIdx.y - (Nreg + U)) * (int)gridDim.z + (int)blockIdx.z) * (int)gridDim.x + (int)blockIdx.x;
        fcd_tally_f_block<4>(tf, lin, ((int)gridDim.y - (Nreg + U)) * (int)gridDim.z * (int)gridDim.x,
                             reinterpret_cast<unsigned long long (*)[3]>(&turn[0][0]));
        return;
    }
    const int b = FB * __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (b >= NBLK) return;
    const int y = (int)blockIdx.y, lane = (int)(threadIdx.x & 63);
    if (y < Nreg) pack_f_item<SQ>(f_state, Nreg, NBLK, C32, mode, f_S, (int)blockIdx.z, y, b, lane, turn[threadIdx.x >> 6]);
    else {
#pragma unroll
        for (int x = 0; x < FB; ++x)
            if (b + x < NBLK) pack_r_item(r_bits, Nreg, U, NBLK, r_S, (int)blockIdx.z, y - Nreg, b + x, lane, pipe);
    }
}

// ---------------------------------------------------------------------------------------------
// The block step kernel.  Launch s = 0 .. NBLK carries two kinds of independent work (role by blockIdx.x):
//   D(s-1)  the in-order part of block s-1: for each (patient, chain word) walk its 16 regions, drawing r_n;
//   P(s)    the panel sums of block s over every block EXCEPT s-1 and s: what P(s) reads (new r below block s-1,
//           old r above block s) is final before D(s-1) starts, so the two run side by side; D(s) adds the terms
//           against block s-1 (just redrawn) and against its own block itself.  P(s) also makes the thresholds
//           (Philox + logit) of block s, consumed by D(s) in the next launch.
// The few D workgroups come first in the grid and finish before the panel workgroups that start beside them, so a
// launch lasts as long as its panel part -- the in-order part costs no time of its own.
// Both roles use workgroups of (chain words per group) waves and the same LDS size; <= 64 VGPRs, two per CU.
//
// Role P: what limits it is the number of wave-wide LDS reads (one per gathered value whatever the number of
// distinct addresses: profiles/r01_ubench_lds_fp64.txt), so the tile holds PAIR records: for the pair of regions
// (m, m+1) and patient u, all 9 x 4 sums
//   pr[q = 3k + k'][tt = t + 2t'] = lMd[u][n][m][k][t] + lMd[u][n][m+1][k'][t']              (288 bytes)
// built in LDS from the two single rows while staging.  One 8-byte LDS read and one fp64 add then cover
// TWO regions; the address is  q*32 + tt*8  from the packed f / r words (2 integer ops per term + 2 per pair).
// LDS: pairs [8*NBLK][UB][36] doubles + singles [UB][16*NBLK*6] doubles (scratch, zero beyond Nreg).
// ---------------------------------------------------------------------------------------------
struct r_step_args {
    const double *lMd, *hyper;
    const uint2 *f_S;       // f pair bytes (q << 2)
    const uint2 *r_S;       // r bytes (one per pair of regions) before the pass (pack_r): what the blocks above the current one still hold
    uint2 *r_Sn;            // r bytes redrawn in this pass: each written once (by D), read only afterwards -> plain cached loads are safe
    uint64_t *r_bits;
    double *Pbuf[2];        // e = (dpi + panel sum) - threshold: P(s) writes [s & 1], D(s) reads it
    uint32_t *flags;        // pipelined form: one mark per (chain word, patient, block); else nullptr
    int Nreg, U, NBLK, GW;
    int u_lo, u_n;          // the patients this launch serves: [u_lo, u_lo + u_n)  (patients are independent given f: a pass
                            // may run as two half-passes on two streams, see fcd_gibbs_r_step_sq)
    int wpb, nWG;           // chain words per workgroup, groups of chain words
    int s, nD, nP;          // step-per-launch form: the step and the number of workgroups per role
    int ncu, npad;          // ... CUs of the device; empty workgroups at [ncu, ncu + npad) (beside the D workgroups)
    uint32_t chain0, sweep;
    uint64_t seed;
    double tol;             // |v| below this: the draw is re-decided with the exact threshold (>= FCD_LOGIT_FAST_ERR)
    int poll_limit;         // pipelined form: polls before a wait is given up (R_POLL_LIMIT; smaller only through the test hook)
    int withhold;           // TEST HOOK (knob r_withhold): the in-order role never sets its marks
    int dsplit;             // pipelined form: two in-order workgroups per patient (8 chain words each, two CUs)
};

// agent-scope (memory-side) access to what crosses workgroups inside the pipelined launch; plain otherwise
template <bool COH>
__device__ __forceinline__ double ld_d(const double *p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH>
__device__ __forceinline__ void st_d(double *p, double v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

#ifndef FCD_PGRP1
#define FCD_PGRP1 2
#endif
constexpr int P_GRP_2 = 2;           // blocks of 16 regions whose state words are prefetched together: 12 VGPRs a group at 2 patients
constexpr int P_GRP_1 = FCD_PGRP1;   // ... with ONE patient per panel workgroup (8 VGPRs a group of two)
                           // (measured at cfg3: 1 -> 326 us per pass, 2 -> 316, 3 -> 322, 4 -> 352: register pressure)
// role D (doubles): compact = 16 waves; (D_RECS_T - D_SAFE) * 9 rows of entries <= 1024 threads
constexpr int D_LDS_COMPACT = (R_NB * (R_NB / 2) + 104) * 36 + R_NB * R_NB * 6;
constexpr int D_LDS_SPREAD = 2 * R_NB * (R_NB / 2) * 36 + 2 * R_NB * R_NB * 6;

// st = step (block of rows), (row, uc, wg) = region of the block, chunk of patients, group of chain words.
template <int UB>
__device__ __forceinline__ void r_role_panel(const r_step_args &a, int st, int row, int uc, int wg, double *smem) {
    constexpr int P_GRP = UB == 1 ? P_GRP_1 : P_GRP_2;
    const int Nreg = a.Nreg, U = a.U, NBLK = a.NBLK;
    const int x0 = st > 0 ? st - 1 : 0, x1 = st + 1;       // blocks left out of the sums
    const int n_pairs = NBLK * (R_NB / 2);
    double *pairs = smem;                                  // [n_pairs][UB][36]
    double *single = smem + (size_t)n_pairs * UB * 36;     // [UB][16 NBLK regions * 6], zero beyond Nreg
    const int n = st * R_NB + row;
    const int u0 = a.u_lo + uc * UB;
    const int nu = (a.u_lo + a.u_n - u0 < UB) ? (a.u_lo + a.u_n - u0) : UB;
    if (FCD_ABL(1, 5)) return;            // ablation: empty role
    [[maybe_unused]] const int trec = st * 1024 + (int)blockIdx.x;
    FCD_TRACE(trec, 0);
    FCD_TRACE_VAL(trec, 6, 1);
    FCD_TRACE_VAL(trec, 7, (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff));
    {
        // rows padded with zero records to a whole number of blocks: the pair build below needs no guards.  Two loads per
        // thread and turn, both in flight before the first is stored (addresses clamped, zeros selected afterwards: a
        // guarded load would make every turn wait for its own trip to memory)
        const int row_d2 = Nreg * 3, pad_d2 = NBLK * R_NB * 3, total = UB * pad_d2;
        double2 *dst = reinterpret_cast<double2 *>(single);
        const double2 *rowp[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int us = u < nu ? u : nu - 1;            // tail chunk: replicate the last patient (never stored)
            rowp[u] = reinterpret_cast<const double2 *>(a.lMd + ((int64_t)(u0 + us) * Nreg + n) * Nreg * 6);
        }
        constexpr int SU = 2;
        for (int it0 = threadIdx.x; it0 < total; it0 += SU * blockDim.x) {
            double2 v[SU];
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                const int it = it0 + j * (int)blockDim.x;
                const int itc = it < total ? it : total - 1;
                const int u = itc / pad_d2, i = itc - u * pad_d2;
                const double2 *src = rowp[0];
#pragma unroll
                for (int uu = 1; uu < UB; ++uu)
                    if (u == uu) src = rowp[uu];
                const double2 x = src[i < row_d2 ? i : 0];
                v[j] = i < row_d2 ? x : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                const int it = it0 + j * (int)blockDim.x;
                if (it < total) dst[it] = v[j];
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(wg * a.wpb + (threadIdx.x >> 6)));
    const bool live = w < a.GW;
    // The draw of (n, u) is  r = 1  <=>  logit(x) < ln(pi/(1-pi)) + d.  The threshold logit(x) depends on the counter
    // RNG only: it is made here (one Philox block per region and PAIR of patients = this workgroup's two patients),
    // while the rows above are on their way from memory, and leaves with the panel sum as  e = (dpi + sum) - logit~(x);
    // D(st) adds its terms and tests the sign.  logit~ is the 8-instruction fp32 form: D re-decides with the exact
    // logit whenever the sum comes within a.tol of zero (a few draws in 10^5), so every outcome is the exact one's.
    const double dpi = a.hyper[FCD_H_LNPI1] - a.hyper[FCD_H_LNPI0];
    double th[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) th[u] = 0.0;
    if (live && !FCD_ABL(2, 1)) {
        const uint32_t chain = a.chain0 + (uint32_t)w * 64u + lane;
        fcd_u4 x = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int uu = u0 + u;
            if (u == 0 || (uu & 1) == 0)
                x = fcd_philox((uint32_t)(n * ((U + 1) >> 1) + (uu >> 1)), chain, a.sweep, FCD_KIND_R, (uint32_t)a.seed,
                               (uint32_t)(a.seed >> 32));
            th[u] = fcd_logit_fast((uu & 1) ? fcd_u53(x.z, x.w) : fcd_u53(x.x, x.y));
        }
    }
    // wave-uniform bases (scalar registers) + unsigned 32-bit lane offsets: no per-lane 64-bit pointers
    const uint32_t ulane = (uint32_t)lane;
    const uint2 *__restrict__ fr = a.f_S + ((int64_t)(live ? w : 0) * Nreg + n) * NBLK * 64;
    // r bytes: blocks above the current one from the array made before the pass, blocks below from the redrawn one
    // (blocks st-1 and st are loaded with the rest but not used: they come from the old array too, so that no line of
    // the redrawn array is touched -- and cached -- before it is final)
    int64_t rt[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int uu = u < nu ? u : nu - 1;
        rt[u] = ((int64_t)(live ? w : 0) * U + u0 + uu) * NBLK * 64;
    }
    const int64_t redrawn = a.r_Sn - a.r_S;          // element distance between the two arrays (wave-uniform select below)
    auto rword = [&](int u, int b) -> uint2 {
        const uint2 *base = a.r_S + (rt[u] + b * 64 + (b >= st - 1 ? (int64_t)0 : redrawn));
        return base[ulane];
    };
    // state words of the first group of blocks: issued before the barriers, their latency hides behind the staging
    uint2 fpv[P_GRP], rwv[P_GRP][UB];
#pragma unroll
    for (int g = 0; g < P_GRP; ++g) {
        const int b = (g < NBLK) ? g : NBLK - 1;
        fpv[g] = fr[b * 64 + ulane];
#pragma unroll
        for (int u = 0; u < UB; ++u) rwv[g][u] = rword(u, b);
    }
    __syncthreads();
    FCD_TRACE(trec, 1);
    if (FCD_ABL(1, 4)) return;            // ablation: single rows staged, no pair records
    {
        // pair records: a thread keeps one of the 9 (k, k') rows and makes its four (t, t') entries from two 16-byte
        // reads -- [k][t = 0, 1] of region m and [k'][t' = 0, 1] of region m+1 -- and two 16-byte writes: a third of the
        // LDS instructions of one entry per thread, and two turns through the (pair, patient) list instead of seven
        const int q = threadIdx.x % 9, step = blockDim.x / 9;
        const int k = q / 3, k2 = q - 3 * k;
        if ((int)threadIdx.x < step * 9) {
            const int pad6 = NBLK * R_NB * 6;
            for (int pu = threadIdx.x / 9; pu < n_pairs * UB; pu += step) {
                const double *su = single + (pu % UB) * pad6 + (pu / UB) * 12;      // pair (m, m+1), m = 2 (pu / UB)
                const double2 a2 = *reinterpret_cast<const double2 *>(su + 2 * k);          // region m:   t  = 0, 1
                const double2 b2 = *reinterpret_cast<const double2 *>(su + 6 + 2 * k2);     // region m+1: t' = 0, 1
                double2 *dst = reinterpret_cast<double2 *>(pairs + pu * 36 + q * 4);        // tt = t + 2 t'
                dst[0] = make_double2(a2.x + b2.x, a2.y + b2.x);
                dst[1] = make_double2(a2.x + b2.y, a2.y + b2.y);
            }
        }
        __syncthreads();
    }
    FCD_TRACE(trec, 2);
    if (!live) return;
    // LDS byte offset of the tile: reads go through an LDS-space pointer so that (block base + pair, patient offset)
    // becomes scalar base + instruction immediate
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const uint32_t pb_off = lds_base0(pairs);
    double d[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) d[u] = 0.0;
    constexpr uint32_t REC = UB * 288u;   // bytes per pair of regions in the tile
    if (FCD_ABL(1, 3)) return;            // ablation: staging only

    // Blocks of 16 regions in groups of P_GRP: the state words of the NEXT group are requested before the
    // current group's terms run, so no global latency sits on the loop.
#ifdef FCD_ABLATE
    long long tr_wait = 0, tr_terms = 0, tr_issue = 0;           // (diagnostic build: where the loop's time goes, wave 0)
#endif
    for (int bg = 0; bg < NBLK; bg += P_GRP) {
#ifdef FCD_ABLATE
        const long long tc0 = clock64();
#endif
        // (always loaded, from a clamped block index: a guard around each load turns into a branch and a wait per word)
        uint2 fpn[P_GRP], rwn[P_GRP][UB];
#pragma unroll
        for (int g = 0; g < P_GRP; ++g) {
            const int b = (bg + P_GRP + g < NBLK) ? bg + P_GRP + g : NBLK - 1;
            fpn[g] = fr[b * 64 + ulane];
#pragma unroll
            for (int u = 0; u < UB; ++u) rwn[g][u] = rword(u, b);
        }
#ifdef FCD_ABLATE
        const long long tc1 = clock64();                          // the next group's loads are issued
#pragma unroll
        for (int g = 0; g < P_GRP; ++g) {
            asm volatile("" ::"v"(fpv[g].x), "v"(fpv[g].y));
#pragma unroll
            for (int u = 0; u < UB; ++u) asm volatile("" ::"v"(rwv[g][u].x), "v"(rwv[g][u].y));
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_GRP * (1 + UB)) : "memory");          // (this group's words have landed; the next group's stay in flight)
        const long long tc2 = clock64();
#endif
#pragma unroll
        for (int g = 0; g < P_GRP; ++g) {
            const int b = bg + g;
            if (b >= NBLK || (b >= x0 && b < x1)) continue;
            if (FCD_ABL(1, 2)) { d[0] += (double)(fpv[g].x + rwv[g][0].y + rwv[g][UB - 1].x); continue; }   // ablation: loads only
            const uint32_t base = pb_off + (uint32_t)b * ((R_NB / 2) * REC);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                // one byte per pair: (q << 2) | tt, i.e. the offset of the term inside its pair record, / 8
                const uint2 z = make_uint2(fpv[g].x | rwv[g][u].x, fpv[g].y | rwv[g][u].y);
                // (the block's eight terms as a tree, one addition into the running sum: a chain of four dependent additions per
                // block instead of eight -- the loop is bound by what ONE wave can have in flight, profiles/r04_trace_r_loop_cfg5.txt)
                double t8[R_NB / 2];
#pragma unroll
                for (int p = 0; p < R_NB / 2; ++p)
                    t8[p] = *(lds_cdouble *)(uintptr_t)(pair_off(z, p) + base + (uint32_t)p * REC + (uint32_t)u * 288u);
                d[u] += ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
            }
        }
#pragma unroll
        for (int g = 0; g < P_GRP; ++g) {
            fpv[g] = fpn[g];
#pragma unroll
            for (int u = 0; u < UB; ++u) rwv[g][u] = rwn[g][u];
        }
#ifdef FCD_ABLATE
        {
            asm volatile("" ::"v"(d[0]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long tc3 = clock64();
            tr_issue += tc1 - tc0;
            tr_wait += tc2 - tc1;
            tr_terms += tc3 - tc2;
        }
#endif
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
        if (u < nu) a.Pbuf[st & 1][(((int64_t)w * U + u0 + u) * R_NB + row) * 64 + ulane] = (dpi + d[u]) - th[u];
#ifdef FCD_ABLATE
    FCD_TRACE_VAL(trec, 4, tr_wait);
    FCD_TRACE_VAL(trec, 5, tr_terms | (tr_issue << 32));
#endif
    FCD_TRACE(trec, 3);
}

// Role D: one workgroup = one patient x one group of chain words; one wave = one (patient, chain word) scan.
// LDS: pair records [2][16][8][36] of the tiles (block b, block b-1) and (block b, block b); the single records are
// staged into the same space first and each thread keeps its sums in registers across a barrier.
// Per region i, in order:  d = panel sum + 8 pair terms against block b-1 + 8 pair terms against the own block
// (the already redrawn bits below i, the old bits above i; the record of (i, i) is zero), compare with the
// threshold.  All reads of a row are independent; only the 1-bit decision links one row to the next.
constexpr int D_RECS_T = R_NB * (R_NB / 2);             // pair records of one tile
#ifndef FCD_PF_E
#define FCD_PF_E 3
#endif
#ifndef FCD_PF_F
#define FCD_PF_F 2
#endif
constexpr int D_SAFE = 104;                             // records of tile 1 that end before single B starts (compact layout)
// b = block, (u, wg) = patient, group of chain words.
__device__ __forceinline__ void r_role_diag(const r_step_args &a, int b, int u, int wg, double *smem) {
    const int Nreg = a.Nreg, U = a.U, NBLK = a.NBLK;
    const int B0 = b * R_NB;
    const int nb = (Nreg - B0 < R_NB) ? (Nreg - B0) : R_NB;
    const bool hasA = b > 0;
    // LDS (doubles): pair records [tile][i][p][36], tile 0 = columns of block b-1, tile 1 = own block; the single
    // records [i][j][6] of both tiles are staged first.  A full workgroup keeps to D_LDS_COMPACT: single A sits where
    // pair tile 1 will go, single B over its last D_RECS_T - D_SAFE records (built from registers across a barrier);
    // a smaller workgroup (fewer chain words) gets the singles behind the pairs.
    const bool compact = blockDim.x == 1024;
    double *pairs = smem;
    double *sA = compact ? smem + D_RECS_T * 36 : smem + 2 * D_RECS_T * 36;
    double *sB = compact ? smem + (D_RECS_T + D_SAFE) * 36 : sA + R_NB * R_NB * 6;
    if (FCD_ABL(2, 5)) return;           // ablation: empty role
    [[maybe_unused]] const int trec = (b + 1) * 1024 + (int)blockIdx.x;
    FCD_TRACE(trec, 0);
    FCD_TRACE_VAL(trec, 6, 2);
    FCD_TRACE_VAL(trec, 7, (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff));
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(wg * a.wpb + (threadIdx.x >> 6)));
    const bool live = w < a.GW;
    const int64_t wu = (int64_t)(live ? w : 0) * U + u;
    // wave-uniform bases (scalar registers) + the lane: no per-lane 64-bit pointers held across the scan
    uint2 *__restrict__ rSn = a.r_Sn + (wu * NBLK + b) * 64;
    const uint2 *__restrict__ frw = a.f_S + (((int64_t)(live ? w : 0) * Nreg + B0) * NBLK + b) * 64;
    const double *__restrict__ Pw = a.Pbuf[b & 1] + (wu * R_NB) * 64;
    // e_i (P(b) wrote them in the previous launch) and the f words are requested PF_E - 1 resp. PF_F - 1 rows ahead of the
    // scan; the first ones leave HERE, before the tiles are staged and built (everything the scan reads was final
    // before the launch).
    const uint32_t ulane = (uint32_t)lane;
    constexpr int PF_E = FCD_PF_E, PF_F = FCD_PF_F;
    uint2 rcur = make_uint2(0u, 0u), rpb = make_uint2(0u, 0u);      // r bytes (one per pair: tt) of the own block (old) / of block b-1 (redrawn)
    double ev[PF_E];
    uint2 fa[PF_F], fb[PF_F];
    {
        rcur = a.r_S[(wu * NBLK + b) * 64 + ulane];
        rpb = (rSn - (hasA ? 64 : 0))[ulane];                       // (block 0: a dummy read; masked where the scan starts)
#pragma unroll
        for (int i = 0; i < PF_E - 1; ++i) ev[i] = Pw[(i < nb ? i : nb - 1) * 64 + ulane];
#pragma unroll
        for (int i = 0; i < PF_F - 1; ++i) {
            const uint2 *fro = frw + (i < nb ? i : nb - 1) * NBLK * 64;
            fb[i] = fro[ulane];
            fa[i] = hasA ? (fro - 64)[ulane] : make_uint2(0u, 0u);
        }
    }
    {
        // single records of both tiles, 16 bytes a piece: (tile, row i, piece c of the row's 768 bytes); two pieces per
        // thread and turn in flight together (clamped addresses, zeros selected afterwards: no load behind a branch)
        const double2 *rowbase = reinterpret_cast<const double2 *>(a.lMd + ((int64_t)u * Nreg + B0) * Nreg * 6) + B0 * 3;
        constexpr int TILE_D2 = R_NB * R_NB * 3;
        double2 *dA = reinterpret_cast<double2 *>(sA), *dB = reinterpret_cast<double2 *>(sB);
        constexpr int SU = 2;
        for (int it0 = threadIdx.x; it0 < 2 * TILE_D2; it0 += SU * blockDim.x) {
            double2 v[SU];
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                const int it = it0 + j * (int)blockDim.x;
                const int itc = it < 2 * TILE_D2 ? it : 2 * TILE_D2 - 1;
                const int tile = itc / TILE_D2, rem = itc - tile * TILE_D2;
                const int i = rem / (R_NB * 3), c = rem - i * (R_NB * 3);
                const bool on = i < nb && (tile == 1 ? c < nb * 3 : hasA);          // beyond Nreg: zero records
                const int64_t off = on ? (int64_t)i * Nreg * 3 + c - (tile == 1 ? 0 : R_NB * 3) : 0;
                const double2 x = rowbase[off];
                v[j] = on ? x : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < SU; ++j) {
                const int it = it0 + j * (int)blockDim.x;
                if (it < 2 * TILE_D2) {
                    const int tile = it / TILE_D2, rem = it - tile * TILE_D2;
                    (tile == 1 ? dB : dA)[rem] = v[j];
                }
            }
        }
    }
    __syncthreads();
    FCD_TRACE(trec, 1);
    {
        // pair records, as in the panel role: a thread keeps one of the 9 (k, k') rows and makes its four (t, t') entries
        // with 16-byte reads and writes; record (i, p) <- singles (i*16 + 2p) * 6
        const int q = threadIdx.x % 9, step = blockDim.x / 9;
        const int k = q / 3, k2 = q - 3 * k;
        const int r0 = threadIdx.x / 9;
        const bool on = (int)threadIdx.x < step * 9;
        auto four = [&](const double *single_tile, int rec, double2 &lo, double2 &hi) {
            const double2 a2 = *reinterpret_cast<const double2 *>(single_tile + rec * 12 + 2 * k);
            const double2 b2 = *reinterpret_cast<const double2 *>(single_tile + rec * 12 + 6 + 2 * k2);
            lo = make_double2(a2.x + b2.x, a2.y + b2.x);
            hi = make_double2(a2.x + b2.y, a2.y + b2.y);
        };
        auto put = [&](int rec_abs, const double2 &lo, const double2 &hi) {
            double2 *dst = reinterpret_cast<double2 *>(pairs + rec_abs * 36 + q * 4);
            dst[0] = lo;
            dst[1] = hi;
        };
        if (on)
            for (int rec = r0; rec < D_RECS_T; rec += step) {
                double2 lo, hi;
                four(sA, rec, lo, hi);
                put(rec, lo, hi);
            }
        __syncthreads();                                  // single A is free: tile 1 may overwrite it
        const int safe = compact ? D_SAFE : D_RECS_T;
        if (on)
            for (int rec = r0; rec < safe; rec += step) {
                double2 lo, hi;
                four(sB, rec, lo, hi);
                put(D_RECS_T + rec, lo, hi);
            }
        if (compact) {                                    // the records single B sits on: one row of entries per thread
            const int rec = D_SAFE + r0;
            const bool mine = on && rec < D_RECS_T;
            double2 lo = make_double2(0.0, 0.0), hi = lo;
            if (mine) four(sB, rec, lo, hi);
            __syncthreads();
            if (mine) put(D_RECS_T + rec, lo, hi);
        }
    }
    __syncthreads();
    FCD_TRACE(trec, 2);
    if (!live || FCD_ABL(2, 3)) return;
    // The scan is the serial chain of the pass (one per patient): its waves go ahead of the panel waves that share the CU.
    __builtin_amdgcn_s_setprio(3);
    // Per region i, in order:  v = e_i (= dpi + panel sum - threshold, from P(b)) + 8 pair terms against block b-1 (its
    // r bits are final) + 8 pair terms against the own block -- redrawn bits below i, old bits above i, the record of
    // (i, i) is zero -- and the sign test.  All 16 reads of a row are independent; only the 1-bit decision links one
    // row to the next.
    if (!hasA) rpb = make_uint2(0u, 0u);
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const uint32_t pb_off = lds_base0(pairs);
    uint32_t fresh = 0;
#pragma unroll
    for (int i = 0; i < R_NB; ++i) {
        if (i < nb) {
            {
                const int ie = i + PF_E - 1, jf = i + PF_F - 1;
                ev[ie % PF_E] = Pw[(ie < nb ? ie : nb - 1) * 64 + ulane];
                const uint2 *fro = frw + (jf < nb ? jf : nb - 1) * NBLK * 64;
                fb[jf % PF_F] = fro[ulane];
                fa[jf % PF_F] = hasA ? (fro - 64)[ulane] : make_uint2(0u, 0u);
            }
            const uint2 fwa = fa[i % PF_F], fwb = fb[i % PF_F];
            double v = ev[i % PF_E];
            if (FCD_ABL(2, 2)) { fresh |= (v + (double)(fwa.x + fwb.y) > 0.0 ? 1u : 0u) << i; continue; }
            // one byte per pair, (q << 2) | tt, OR-ed once per tile; a term = 6 bits of it x 8 + the record's place in the
            // tile (an LDS-space address: scalar base + lane offset + immediate)
            const uint2 za = make_uint2(fwa.x | rpb.x, fwa.y | rpb.y), zb = make_uint2(fwb.x | rcur.x, fwb.y | rcur.y);
            double sa, sb;
            {
                double ta[R_NB / 2];
#pragma unroll
                for (int p = 0; p < R_NB / 2; ++p)
                    ta[p] = *(lds_cdouble *)(uintptr_t)(pair_off6(za, p) + pb_off + (uint32_t)((i * (R_NB / 2) + p) * 288));
                sa = ((ta[0] + ta[1]) + (ta[2] + ta[3])) + ((ta[4] + ta[5]) + (ta[6] + ta[7]));
            }
            {
                double tb[R_NB / 2];
#pragma unroll
                for (int p = 0; p < R_NB / 2; ++p)
                    tb[p] = *(lds_cdouble *)(uintptr_t)(pair_off6(zb, p) + pb_off + (uint32_t)(((R_NB + i) * (R_NB / 2) + p) * 288));
                sb = ((tb[0] + tb[1]) + (tb[2] + tb[3])) + ((tb[4] + tb[5]) + (tb[6] + tb[7]));
            }
            v = (v + sa) + sb;
            if (__ballot(fabs(v) < a.tol) != 0ull) {
                // too close to call with the fast threshold in e_i: put the exact one in its place
                const fcd_u4 x = fcd_philox((uint32_t)((B0 + i) * ((U + 1) >> 1) + (u >> 1)), a.chain0 + (uint32_t)w * 64u + ulane,
                                            a.sweep, FCD_KIND_R, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                const double xx = (u & 1) ? fcd_u53(x.z, x.w) : fcd_u53(x.x, x.y);
                const double corr = fcd_logit_fast(xx) - fcd_logit(xx);
                if (xx > 0.0) v += corr;                    // (x = 0: both thresholds are -inf, v = +inf already)
            }
            const uint32_t t = v > 0.0 ? 1u : 0u;
            fresh |= t << i;
            // region i now carries its new value for the rows below: bit (i & 1) of byte i / 2
            {
                constexpr uint32_t one = 1u;
                const int sh = 8 * ((i >> 1) & 3) + (i & 1);
                if ((i >> 1) < 4) rcur.x = (rcur.x & ~(one << sh)) | (t << sh);
                else rcur.y = (rcur.y & ~(one << sh)) | (t << sh);
            }
        }
    }
    {
        // the same bits, one byte per pair, for the panel role
        rSn[ulane] = spread2(fresh);
    }
#pragma unroll
    for (int i = 0; i < R_NB; ++i) {
        if (i < nb) {
            const uint64_t ball = __ballot((fresh >> i) & 1u);
            if (lane == 0) a.r_bits[((int64_t)w * Nreg + B0 + i) * U + u] = ball;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    FCD_TRACE(trec, 3);
}

// ---------------------------------------------------------------------------------------------
// PIPELINED one-launch form (the default wherever all its workgroups fit the device at once; knob r_path = 3 keeps the
// step-per-launch form).  Same roles, same arithmetic, same chains bit for bit as the step-per-
// launch form, but every workgroup is resident for the whole pass and walks its steps back to back:
//   panel workgroup (row, chunk of UB patients):  P(0), P(1), ... of its region-of-the-block,
//   in-order workgroup (patient):                 D(0), D(1), ... .
// What crosses workgroups carries its own "ready" mark, so there are no counters, no workgroup barriers around the
// hand-over and no read-modify-write atomics:
//   * D(b) writes the redrawn r bytes of its block through to the memory side (agent-scope stores), waits until they
//     are acknowledged and then sets ONE mark word per (chain word, patient, block) (pack_f_kernel cleared the marks
//     before the pass).  A panel wave polls exactly the marks it needs -- its own chain word, its own patients, block
//     st-2 -- and only then reads the bytes, with plain cached loads like every older block: a line of r_Sn is never
//     touched before it is final, so no cache can hold a stale copy of it.
//   * P(st) publishes e = (dpi + panel sum) - threshold with agent-scope stores over a NaN sentinel (R_SENT) that
//     pack_f_kernel put there before the pass; the D wave that owns (patient, chain word) polls the value itself and
//     puts the sentinel back once it has used it (the slot serves block st+2 next).  D orders "sentinels back, r_bits
//     out" before its marked store (vmcnt 0), and P(st+2) writes the slot only after it has seen that mark.
// A panel workgroup requests the table rows of its NEXT step right after it has built the records of the current one,
// makes the thresholds while they fly and parks them in the single-row scratch (free until the next build): the
// staging of a step costs nothing on its own.  Every poll is bounded; a wait that is given up raises the context's
// error word (pinned host memory) and every wave drains without waiting again.
// ---------------------------------------------------------------------------------------------
#ifndef FCD_PIPE_PF_E
#define FCD_PIPE_PF_E 3
#endif
#ifndef FCD_PIPE_PF_F
#define FCD_PIPE_PF_F 2
#endif
// (diagnostic build only, FCD_TRACE_ROW=1: six stamps inside row 8 of every block of the in-order scan, wave 0 --
// profiles/trace_pipe.py; each stamp costs about half a microsecond)
#ifdef FCD_ABLATE
#define PIPE_ROW_USE(x) do { if (i == 8 && FCD_ABL(3, 1)) asm volatile("" ::"v"(x)); } while (0)
#define PIPE_ROW_STAMP(k) do { if (i == 8 && FCD_ABL(3, 1)) { FCD_TRACE((16 + b) * 1024 + (int)blockIdx.x, k); } } while (0)
#else
#define PIPE_ROW_USE(x) do { } while (0)
#define PIPE_ROW_STAMP(k) do { } while (0)
#endif
#ifndef PIPE_SCHED_BARRIER
#define PIPE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int R_PIPE_SLOT_MARGIN = 8;                              // free workgroup slots the pipelined form insists on
constexpr int R_POLL_LIMIT = 1 << 20;                              // x (sleep + one trip to the memory side): about a second

// err: the context's pinned host word.  ok: false once this wave has given up (it then never waits again).
__device__ __forceinline__ void pipe_give_up(volatile unsigned *err, bool &ok) {
    __hip_atomic_store(const_cast<unsigned *>(err), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ok = false;
}
__device__ __forceinline__ void pipe_poll_mark(const uint32_t *mark, int limit, volatile unsigned *err, bool &ok) {
    int spins = 0;
    while (ok) {
        const uint32_t v = __hip_atomic_load(mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (wave-uniform address)
        if (__builtin_amdgcn_readfirstlane((int)v) != 0) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > limit) pipe_give_up(err, ok);
        else if ((spins & 255) == 0 &&
                 __hip_atomic_load(const_cast<unsigned *>(err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
            ok = false;
    }
    asm volatile("" ::: "memory");          // (compiler only: the loads of the bytes stay behind the poll)
}
// The panel value at p is not there yet (some lane still reads the sentinel): poll until it is, or give up.  Out of line
// on purpose -- a loop with a load in it, inlined into the 16-row scan, makes the compiler wait for EVERY load in flight
// (the next rows' requests included) at each row, i.e. one trip to the memory side per row.
__device__ __attribute__((noinline)) double pipe_wait_e(const double *p, int limit, volatile unsigned *err) {
    double e;
    int spins = 0;
    bool ok = true;
    do {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > limit) pipe_give_up(err, ok);
        else if ((spins & 255) == 0 &&
                 __hip_atomic_load(const_cast<unsigned *>(err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
            ok = false;
        e = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while (ok && __ballot((unsigned long long)__double_as_longlong(e) == R_SENT) != 0ull);
    return e;                                   // (still the sentinel in some lane: the wait was given up)
}
__device__ __forceinline__ double pipe_poll_e(const double *p, double first, int limit, volatile unsigned *err, bool &ok) {
    double e = first;
    if (__builtin_expect(ok && __ballot((unsigned long long)__double_as_longlong(e) == R_SENT) != 0ull, 0)) {
        e = pipe_wait_e(p, limit, err);
        if (__ballot((unsigned long long)__double_as_longlong(e) == R_SENT) != 0ull) ok = false;
    }
    return e;
}

// logit~(x) - logit(x) of the draw (idx, chain, sweep): what turns the fast threshold inside e into the exact one.  Out of
// line on purpose: it runs for about one row in a hundred, and inlined into the 16-row scan its fp64 logarithm keeps
// a dozen registers of constants alive across the whole loop (they spill, and every reload waits for the loads in flight).
__device__ __attribute__((noinline)) double pipe_exact_corr(uint32_t idx, uint32_t chain, uint32_t sweep, uint64_t seed, int odd) {
    const fcd_u4 x = fcd_philox(idx, chain, sweep, FCD_KIND_R, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double xx = odd ? fcd_u53(x.z, x.w) : fcd_u53(x.x, x.y);
    return xx > 0.0 ? fcd_logit_fast(xx) - fcd_logit(xx) : 0.0;      // (x = 0: both thresholds are -inf, v = +inf already)
}

// Panel workgroup of the pipelined form: all steps of (row, uc) for the group wg of chain words.
template <int UB>
__device__ __forceinline__ void pipe_panel(const r_step_args &a, int row, int uc, int wg, double *smem, volatile unsigned *err) {
    constexpr int P_GRP = UB == 1 ? P_GRP_1 : P_GRP_2;
    const int Nreg = a.Nreg, U = a.U, NBLK = a.NBLK;
    const int n_pairs = NBLK * (R_NB / 2);
    double *pairs = smem;                                  // [n_pairs][UB][36]
    double *single = smem + (size_t)n_pairs * UB * 36;     // [UB][16 NBLK regions * 6], zero beyond Nreg
    const int u0 = a.u_lo + uc * UB;
    const int nu = (a.u_lo + a.u_n - u0 < UB) ? (a.u_lo + a.u_n - u0) : UB;
    const int lane = threadIdx.x & 63;
    const uint32_t ulane = (uint32_t)lane;
    const int w = __builtin_amdgcn_readfirstlane((int)(wg * a.wpb + (threadIdx.x >> 6)));
    const bool live = w < a.GW;
    const int wl = live ? w : 0;
    const int row_d2 = Nreg * 3, pad_d2 = NBLK * R_NB * 3, total = UB * pad_d2;
    constexpr int SU = 2;                                  // 16-byte pieces of the rows per thread and turn
    bool ok = true;
    int64_t rt[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int uu = u < nu ? u : nu - 1;                // tail chunk: replicate the last patient (never stored)
        rt[u] = ((int64_t)wl * U + u0 + uu) * NBLK * 64;
    }
    const int64_t redrawn = a.r_Sn - a.r_S;
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const uint32_t pb_off = lds_base0(pairs);
    constexpr uint32_t REC = UB * 288u;

    // rows of region n for the UB patients: pieces it0 + j * blockDim of [UB][pad_d2] (clamped loads, zeros beyond Nreg)
    auto load_rows = [&](int n, int it0, double2 (&v)[SU]) {
#pragma unroll
        for (int j = 0; j < SU; ++j) {
            const int it = it0 + j * (int)blockDim.x;
            const int itc = it < total ? it : total - 1;
            const int u = itc / pad_d2, i = itc - u * pad_d2;
            const int us = u < nu ? u : nu - 1;
            const double2 *src = reinterpret_cast<const double2 *>(a.lMd + ((int64_t)(u0 + us) * Nreg + n) * Nreg * 6);
            const double2 x = src[i < row_d2 ? i : 0];
            v[j] = i < row_d2 ? x : make_double2(0.0, 0.0);
        }
    };
    auto store_rows = [&](int it0, const double2 (&v)[SU]) {
        double2 *dst = reinterpret_cast<double2 *>(single);
#pragma unroll
        for (int j = 0; j < SU; ++j) {
            const int it = it0 + j * (int)blockDim.x;
            if (it < total) dst[it] = v[j];
        }
    };
    bool staged = false;                                   // the rows of the step at hand are already in the scratch
    for (int st = 0; st < NBLK; ++st) {
        const int rows = (Nreg - st * R_NB < R_NB) ? (Nreg - st * R_NB) : R_NB;
        if (row >= rows) break;                            // (only the last block can be short)
        const int n = st * R_NB + row;
        [[maybe_unused]] const int trec = st * 1024 + (int)blockIdx.x;
        FCD_TRACE(trec, 0);
        FCD_TRACE_VAL(trec, 6, 1);
        FCD_TRACE_VAL(trec, 7, (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff));
        if (!staged) {
            for (int it0 = threadIdx.x; it0 < total; it0 += SU * blockDim.x) {
                double2 v[SU];
                load_rows(n, it0, v);
                store_rows(it0, v);
            }
        }
        __syncthreads();                                   // rows in place; every wave is done with the previous records
        {
            const int q = threadIdx.x % 9, step = blockDim.x / 9;
            const int k = q / 3, k2 = q - 3 * k;
            if ((int)threadIdx.x < step * 9) {
                const int pad6 = NBLK * R_NB * 6;
                for (int pu = threadIdx.x / 9; pu < n_pairs * UB; pu += step) {
                    const double *su = single + (pu % UB) * pad6 + (pu / UB) * 12;      // pair (m, m+1), m = 2 (pu / UB)
                    const double2 a2 = *reinterpret_cast<const double2 *>(su + 2 * k);
                    const double2 b2 = *reinterpret_cast<const double2 *>(su + 6 + 2 * k2);
                    double2 *dst = reinterpret_cast<double2 *>(pairs + pu * 36 + q * 4);
                    dst[0] = make_double2(a2.x + b2.x, a2.y + b2.x);
                    dst[1] = make_double2(a2.x + b2.y, a2.y + b2.y);
                }
            }
        }
        __syncthreads();                                   // records in place; the scratch is free
        FCD_TRACE(trec, 1);
        // the NEXT step's rows: requested now, parked in the scratch after the thresholds are made (one turn of the
        // piece loop covers every shape whose rows fit the LDS at all: total <= 2 * blockDim pieces ... else a second turn)
        const int rows_next = st + 1 < NBLK ? ((Nreg - (st + 1) * R_NB < R_NB) ? (Nreg - (st + 1) * R_NB) : R_NB) : 0;
        const bool more = row < rows_next;
        double2 nv[SU];
        unsigned txp = threadIdx.x;                           // (opaque copy: the piece index is made here, not carried -- spilled -- across the step)
        asm volatile("" : "+v"(txp));
        if (more) load_rows(n + R_NB, (int)txp, nv);
        // The marks of block st-2 (one per patient) are ASKED FOR here, a whole threshold computation and most of the sums
        // before they are needed: in the usual case they have long been set, and the two memory-side trips of the poll (one
        // per patient, one after the other: ~1.4 us at the end of every step) are off the step.  A mark only ever goes
        // 0 -> 1 inside a pass, so an early 1 is final; an early 0 falls back to the poll where the bytes are needed.
        uint32_t mk[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            mk[u] = 1u;
            if (st >= 2) {
                const int uu = u < nu ? u : nu - 1;
                mk[u] = __hip_atomic_load(a.flags + ((int64_t)wl * U + u0 + uu) * NBLK + (st - 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        double th[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) th[u] = 0.0;
        if (live) {
            const uint32_t chain = a.chain0 + (uint32_t)w * 64u + lane;
            fcd_u4 x = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int uu = u0 + u;
                if (u == 0 || (uu & 1) == 0)
                    x = fcd_philox((uint32_t)(n * ((U + 1) >> 1) + (uu >> 1)), chain, a.sweep, FCD_KIND_R, (uint32_t)a.seed,
                                   (uint32_t)(a.seed >> 32));
                th[u] = fcd_logit_fast((uu & 1) ? fcd_u53(x.z, x.w) : fcd_u53(x.x, x.y));
            }
        }
        if (more) {
            store_rows(threadIdx.x, nv);
            for (int it0 = threadIdx.x + SU * blockDim.x; it0 < total; it0 += SU * blockDim.x) {     // (rows longer than one turn)
                double2 v[SU];
                load_rows(n + R_NB, it0, v);
                store_rows(it0, v);
            }
        }
        staged = more;
        bool seen[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) seen[u] = __builtin_amdgcn_readfirstlane((int)mk[u]) != 0;
        if (!live) continue;
        FCD_TRACE(trec, 2);
        // The sums run over the blocks ABOVE the current one first (their r bytes are the old ones, made before the pass),
        // then over blocks 0 .. st-2 in ascending order: the only bytes that may not be there yet -- block st-2, being
        // redrawn by D(st-2) while this workgroup worked on step st-1 -- are asked for last, and the wave looks for
        // their marks just before it requests them (group by group, one group ahead of the terms).
        const int nUp = NBLK - 1 - st;                     // blocks st+1 .. NBLK-1
        const int nLo = st >= 2 ? st - 1 : 0;              // blocks 0 .. st-2
        const int NV = nUp + nLo;
        const uint2 *__restrict__ fr = a.f_S + ((int64_t)wl * Nreg + n) * NBLK * 64;
        auto blk = [&](int j) -> int {
            const int jc = j < NV ? j : NV - 1;
            return jc < nUp ? st + 1 + jc : jc - nUp;
        };
        auto rword = [&](int u, int b) -> uint2 {
            const uint2 *base = a.r_S + (rt[u] + b * 64 + (b >= st - 1 ? (int64_t)0 : redrawn));
            return base[ulane];
        };
        bool polled = st < 2;
        auto poll_before = [&](int jlast) {                // the last entry of the list is block st-2
            if (!polled && jlast >= NV - 1) {
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (u < nu && !seen[u]) pipe_poll_mark(a.flags + ((int64_t)wl * U + u0 + u) * NBLK + (st - 2), a.poll_limit, err, ok);
                polled = true;
                FCD_TRACE(trec, 3);
            }
        };
        double d[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) d[u] = 0.0;
        if (FCD_ABL(1, 2)) {                    // (ablation: no terms; the hand-over stays)
            poll_before(NV - 1);
        } else if (NV > 0) {
            uint2 fpv[P_GRP], rwv[P_GRP][UB];
            poll_before(P_GRP - 1);
#pragma unroll
            for (int g = 0; g < P_GRP; ++g) {
                const int b = blk(g);
                fpv[g] = fr[b * 64 + ulane];
#pragma unroll
                for (int u = 0; u < UB; ++u) rwv[g][u] = rword(u, b);
            }
            for (int jg = 0; jg < NV; jg += P_GRP) {
                uint2 fpn[P_GRP], rwn[P_GRP][UB];
                poll_before(jg + 2 * P_GRP - 1);
#pragma unroll
                for (int g = 0; g < P_GRP; ++g) {
                    const int b = blk(jg + P_GRP + g);
                    fpn[g] = fr[b * 64 + ulane];
#pragma unroll
                    for (int u = 0; u < UB; ++u) rwn[g][u] = rword(u, b);
                }
#pragma unroll
                for (int g = 0; g < P_GRP; ++g) {
                    if (jg + g >= NV) continue;
                    const uint32_t base = pb_off + (uint32_t)blk(jg + g) * ((R_NB / 2) * REC);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const uint2 z = make_uint2(fpv[g].x | rwv[g][u].x, fpv[g].y | rwv[g][u].y);
                        // (two trees of four per block -- see r_role_panel; a tree of eight costs this kernel eight registers it has not)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            double t4[4];
#pragma unroll
                            for (int p = 0; p < 4; ++p)
                                t4[p] = *(lds_cdouble *)(uintptr_t)(pair_off6(z, 4 * h + p) + base + (uint32_t)(4 * h + p) * REC + (uint32_t)u * 288u);
                            d[u] += (t4[0] + t4[1]) + (t4[2] + t4[3]);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < P_GRP; ++g) {
                    fpv[g] = fpn[g];
#pragma unroll
                    for (int u = 0; u < UB; ++u) rwv[g][u] = rwn[g][u];
                }
            }
        }
        // (ln pi - ln(1 - pi) read HERE, behind an index the compiler cannot see through: lifted out of the step loop it is one
        //  more value alive across the sums -- spilled, and fetched back from scratch memory right in front of this store)
        int hz = 0;
        asm volatile("" : "+s"(hz));
        const double dpi = a.hyper[FCD_H_LNPI1 + hz] - a.hyper[FCD_H_LNPI0 + hz];
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (u < nu) st_d<true>(a.Pbuf[st & 1] + (((int64_t)w * U + u0 + u) * R_NB + row) * 64 + ulane, (dpi + d[u]) - th[u]);
        FCD_TRACE(trec, 4);
    }
}

// In-order workgroup of the pipelined form: all blocks of patient u for the group wg of chain words.
// half < 0: the workgroup scans all its chain words; else only the 8 words of that half (waves 8 half .. 8 half + 7) --
// the patient's other workgroup, on another CU, scans the rest; every wave still helps to stage and build the tiles.
__device__ __forceinline__ void pipe_diag(const r_step_args &a, int u, int wg, int half, double *smem, volatile unsigned *err) {
    const int Nreg = a.Nreg, U = a.U, NBLK = a.NBLK;
    const bool compact = blockDim.x == 1024;
    double *pairs = smem;
    double *sA = compact ? smem + D_RECS_T * 36 : smem + 2 * D_RECS_T * 36;
    double *sB = compact ? smem + (D_RECS_T + D_SAFE) * 36 : sA + R_NB * R_NB * 6;
    const int lane = threadIdx.x & 63;
    const uint32_t ulane = (uint32_t)lane;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = wg * a.wpb + wave;
    const bool live = w < a.GW && (half < 0 || (wave >> 3) == half);
    const int64_t wu = (int64_t)(live ? w : 0) * U + u;
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const uint32_t pb_off = lds_base0(pairs);
    bool ok = true;
    [[maybe_unused]] const unsigned tr0 = half == 1 ? 512u : 0u;      // (diagnostic build: the thread that stamps -- a live one)
    uint2 rpb = make_uint2(0u, 0u);                       // r bytes of block b-1 as this wave redrew them
    for (int b = 0; b < NBLK; ++b) {
        const int B0 = b * R_NB;
        const int nb = (Nreg - B0 < R_NB) ? (Nreg - B0) : R_NB;
        const bool hasA = b > 0;
        // (the thread index behind an opaque copy: the staging / build addresses derived from it are made anew in every
        //  block -- a dozen instructions -- instead of being lifted out of the loop, spilled for want of registers and
        //  fetched back from scratch memory, one trip each, in the middle of every block's build)
        unsigned tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        // (block 0 has no tile A: its records are zero whatever the bytes say, so the f words of tile A are then read from
        // the block itself -- a valid address, no branch around the load)
        const int a_back = hasA ? 64 : 0;
        [[maybe_unused]] const int trec = b * 1024 + (int)blockIdx.x;
        FCD_TRACE_AT(tr0, trec, 0);
        FCD_TRACE_VAL_AT(tr0, trec, 6, 2);
        FCD_TRACE_VAL_AT(tr0, trec, 7, (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff));
        const uint2 *__restrict__ frw = a.f_S + (((int64_t)(live ? w : 0) * Nreg + B0) * NBLK + b) * 64;
        const double *__restrict__ Pw = a.Pbuf[b & 1] + (wu * R_NB) * 64;
        // e_i are asked for PF_E - 1 rows ahead (agent-scope loads: a trip to the memory side), the f words PF_F - 1 rows
        // ahead (cached), and within a row the f words go first: loads return in order, so a row that waits for its f words
        // must not find a younger e request in front of them.  (At one row ahead each of the 16 rows waited for a whole
        // trip: 0.67 us per row with nothing else to do.)
        constexpr int PF_E = FCD_PIPE_PF_E, PF_F = FCD_PIPE_PF_F;
        double ev[PF_E];
        uint2 fa[PF_F], fb[PF_F];
        // the block's old r bytes (pack_f_kernel made them before the pass) and the first rows' e and f words: asked for
        // BEFORE the tiles, so that they have landed by the time the staged rows have (no load is left in flight across
        // the build, where a wait for anything is a wait for everything older)
        uint2 rcur = a.r_S[(wu * NBLK + b) * 64 + ulane];
#pragma unroll
        for (int i = 0; i < PF_F - 1; ++i) {
            const uint2 *fro = frw + (i < nb ? i : nb - 1) * NBLK * 64;
            fb[i] = fro[ulane];
            fa[i] = (fro - a_back)[ulane];
        }
#pragma unroll
        for (int i = 0; i < PF_E - 1; ++i) ev[i] = ld_d<true>(Pw + (i < nb ? i : nb - 1) * 64 + ulane);
        __syncthreads();                                   // every wave is done with the previous block's records
        FCD_TRACE_AT(tr0, trec, 3);
        {
            const double2 *rowbase = reinterpret_cast<const double2 *>(a.lMd + ((int64_t)u * Nreg + B0) * Nreg * 6) + B0 * 3;
            constexpr int TILE_D2 = R_NB * R_NB * 3;
            double2 *dA = reinterpret_cast<double2 *>(sA), *dB = reinterpret_cast<double2 *>(sB);
            constexpr int SU = 2;
            for (int it0 = (int)tx; it0 < 2 * TILE_D2; it0 += SU * blockDim.x) {
                double2 v[SU];
#pragma unroll
                for (int j = 0; j < SU; ++j) {
                    const int it = it0 + j * (int)blockDim.x;
                    const int itc = it < 2 * TILE_D2 ? it : 2 * TILE_D2 - 1;
                    const int tile = itc / TILE_D2, rem = itc - tile * TILE_D2;
                    const int i = rem / (R_NB * 3), c = rem - i * (R_NB * 3);
                    const bool on = i < nb && (tile == 1 ? c < nb * 3 : hasA);
                    const int64_t off = on ? (int64_t)i * Nreg * 3 + c - (tile == 1 ? 0 : R_NB * 3) : 0;
                    const double2 x = rowbase[off];
                    v[j] = on ? x : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int j = 0; j < SU; ++j) {
                    const int it = it0 + j * (int)blockDim.x;
                    if (it < 2 * TILE_D2) {
                        const int tile = it / TILE_D2, rem = it - tile * TILE_D2;
                        (tile == 1 ? dB : dA)[rem] = v[j];
                    }
                }
            }
        }
        __syncthreads();
        FCD_TRACE_VAL_AT(tr0, trec, 5, wall_clock64());
        {
            const int q = tx % 9, step = blockDim.x / 9;
            const int k = q / 3, k2 = q - 3 * k;
            const int r0 = tx / 9;
            const bool on = (int)tx < step * 9;
            auto four = [&](const double *single_tile, int rec, double2 &lo, double2 &hi) {
                const double2 a2 = *reinterpret_cast<const double2 *>(single_tile + rec * 12 + 2 * k);
                const double2 b2 = *reinterpret_cast<const double2 *>(single_tile + rec * 12 + 6 + 2 * k2);
                lo = make_double2(a2.x + b2.x, a2.y + b2.x);
                hi = make_double2(a2.x + b2.y, a2.y + b2.y);
            };
            auto put = [&](int rec_abs, const double2 &lo, const double2 &hi) {
                double2 *dst = reinterpret_cast<double2 *>(pairs + rec_abs * 36 + q * 4);
                dst[0] = lo;
                dst[1] = hi;
            };
            if (on)
                for (int rec = r0; rec < D_RECS_T; rec += step) {
                    double2 lo, hi;
                    four(sA, rec, lo, hi);
                    put(rec, lo, hi);
                }
            __syncthreads();                                  // single A is free: tile 1 may overwrite it
            const int safe = compact ? D_SAFE : D_RECS_T;
            if (on)
                for (int rec = r0; rec < safe; rec += step) {
                    double2 lo, hi;
                    four(sB, rec, lo, hi);
                    put(D_RECS_T + rec, lo, hi);
                }
            if (compact) {
                const int rec = D_SAFE + r0;
                const bool mine = on && rec < D_RECS_T;
                double2 lo = make_double2(0.0, 0.0), hi = lo;
                if (mine) four(sB, rec, lo, hi);
                __syncthreads();
                if (mine) put(D_RECS_T + rec, lo, hi);
            }
        }
        __syncthreads();
        FCD_TRACE_AT(tr0, trec, 1);
        if (!live) continue;
        __builtin_amdgcn_s_setprio(3);
        uint32_t fresh = 0;
        // The scan, software-pipelined.  Of the 16 terms of row i exactly ONE needs the decision of row i - 1: pair
        // pl = (i - 1) >> 1 of the own block (tile B).  Everything else -- e_i, tile A, the other seven pairs of tile B,
        // summed as far as the tree (t0+t1)+(t2+t3) + (t4+t5)+(t6+t7) allows without that leaf -- is the row's EARLY part
        // and is issued while the late term of the row BEFORE is in flight: the dependent chain of a row is one LDS
        // read and four additions instead of the whole row.  (Same tree, same sums: additions commute.)
        int nbv = nb;          // (nb behind an opaque copy: see the loop below)
        auto term = [&](uint32_t zw, int p, uint32_t rbase) -> double {
            const int sft = 8 * (p & 3);
            const uint32_t off = sft == 0 ? (zw << 3) & 0x1F8u : (zw >> (sft - 3)) & 0x1F8u;
            return *(lds_cdouble *)(uintptr_t)(off + rbase + (uint32_t)(p * 288));
        };
        auto four = [&](const uint2 &z, uint32_t rbase, int p0) -> double {
            double t[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) t[p] = term(p0 < 4 ? z.x : z.y, p0 + p, rbase);
            return (t[0] + t[1]) + (t[2] + t[3]);
        };
        auto early = [&](int i, double &vsa, double &Qo, double &Po, double &ts, uint32_t &fwl) {
            {   // the requests for the rows ahead
                const int ie = i + PF_E - 1, jf = i + PF_F - 1;
                const uint2 *fro = frw + (jf < nbv ? jf : nbv - 1) * NBLK * 64;
                fb[jf % PF_F] = fro[ulane];
                fa[jf % PF_F] = (fro - a_back)[ulane];
                ev[ie % PF_E] = ld_d<true>(Pw + (ie < nbv ? ie : nbv - 1) * 64 + ulane);
            }
            const uint2 fwa = fa[i % PF_F], fwb = fb[i % PF_F];
            // one byte per pair: (q << 2) | tt -- tile A against block b-1 (this wave's own redrawn bytes), tile B against
            // the own block (redrawn below i, old above i; the record of (i, i) is zero)
            const uint2 za = make_uint2(fwa.x | rpb.x, fwa.y | rpb.y), zb = make_uint2(fwb.x | rcur.x, fwb.y | rcur.y);
            const uint32_t ra = pb_off + (uint32_t)(i * (R_NB / 2) * 288), rb = pb_off + (uint32_t)((R_NB + i) * (R_NB / 2) * 288);
            const int pl = i > 0 ? (i - 1) >> 1 : 0, ql = pl >> 2, hl = (pl >> 1) & 1;
            // eight reads in flight, then seven: a wave that reads four at a time spends three quarters of a row waiting for
            // the LDS, and there are only two to four such waves per SIMD to fill the gaps
            double ta[8], tb[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) ta[p] = term(p < 4 ? za.x : za.y, p, ra);
            vsa = ev[i % PF_E] + (((ta[0] + ta[1]) + (ta[2] + ta[3])) + ((ta[4] + ta[5]) + (ta[6] + ta[7])));
            PIPE_SCHED_BARRIER();
#pragma unroll
            for (int p = 0; p < 8; ++p) tb[p] = p == pl ? 0.0 : term(p < 4 ? zb.x : zb.y, p, rb);
            const int qo = 4 * (1 - ql), po = 4 * ql + 2 * (1 - hl);
            Qo = (tb[qo] + tb[qo + 1]) + (tb[qo + 2] + tb[qo + 3]);
            Po = tb[po] + tb[po + 1];
            ts = tb[pl ^ 1];
            fwl = pl < 4 ? fwb.x : fwb.y;
        };
        double c_vsa, c_Qo, c_Po, c_ts;
        uint32_t c_fwl;
        early(0, c_vsa, c_Qo, c_Po, c_ts, c_fwl);
        // A row too close to call with the fast threshold inside e_i needs the exact one (pipe_exact_corr: a call, rare).
        // It leaves the unrolled rows for the one call site below and comes back in at the next row: the rows themselves
        // hold no call.
        int i0 = 0;
        for (;;) {
            int stop = -1;
            double vstop = 0.0;
            // The rows sit in a loop now, and the compiler would lift the 32 row addresses out of it (64 registers, spilled):
            // they are made to depend on a value it cannot see through.
            asm volatile("" : "+s"(nbv));
#pragma unroll
            for (int i = 0; i < R_NB; ++i) {
                if (i >= i0 && i < nb) {
                    const int pl = i > 0 ? (i - 1) >> 1 : 0;
                    // the late term: rcur now carries the decision of row i - 1
                    const double tl = term(c_fwl | (pl < 4 ? rcur.x : rcur.y), pl, pb_off + (uint32_t)((R_NB + i) * (R_NB / 2) * 288));
                    PIPE_SCHED_BARRIER();
                    double n_vsa = 0.0, n_Qo = 0.0, n_Po = 0.0, n_ts = 0.0;
                    uint32_t n_fwl = 0u;
                    if (i + 1 < R_NB && i + 1 < nb) early(i + 1, n_vsa, n_Qo, n_Po, n_ts, n_fwl);
                    PIPE_SCHED_BARRIER();
                    const double v = c_vsa + (((tl + c_ts) + c_Po) + c_Qo);
                    c_vsa = n_vsa; c_Qo = n_Qo; c_Po = n_Po; c_ts = n_ts; c_fwl = n_fwl;
                    // not clearly one side of zero in some lane: too close to call with the fast threshold inside e_i -- or
                    // not a number, e_i was still the sentinel when it was asked for
                    if (__builtin_expect(__ballot(!(fabs(v) >= a.tol)) != 0ull, 0)) {
                        stop = i;
                        vstop = v;
                        break;
                    }
                    const uint32_t t = v > 0.0 ? 1u : 0u;
                    fresh |= t << i;
                    // region i now carries its new value for the rows below: bit (i & 1) of byte i / 2
                    constexpr uint32_t one = 1u;
                    const int sh = 8 * ((i >> 1) & 3) + (i & 1);
                    if ((i >> 1) < 4) rcur.x = (rcur.x & ~(one << sh)) | (t << sh);
                    else rcur.y = (rcur.y & ~(one << sh)) | (t << sh);
                    PIPE_SCHED_BARRIER();
                }
            }
            if (__builtin_expect(stop < 0, 1)) break;
            if (__ballot(vstop != vstop) != 0ull) {
                // the panels are behind: wait for e of this row (bounded) and sum the row again, from memory, the plain way
                // (the same tree of additions)
                double e = ld_d<true>(Pw + stop * 64 + ulane);
                if (ok && __ballot((unsigned long long)__double_as_longlong(e) == R_SENT) != 0ull) {
                    e = pipe_wait_e(Pw + stop * 64 + ulane, a.poll_limit, err);
                    if (__ballot((unsigned long long)__double_as_longlong(e) == R_SENT) != 0ull) ok = false;
                }
                const uint2 *fro = frw + stop * NBLK * 64;
                const uint2 fwb = fro[ulane], fwa = (fro - a_back)[ulane];
                const uint2 za = make_uint2(fwa.x | rpb.x, fwa.y | rpb.y), zb = make_uint2(fwb.x | rcur.x, fwb.y | rcur.y);
                const uint32_t ra = pb_off + (uint32_t)(stop * (R_NB / 2) * 288), rb = pb_off + (uint32_t)((R_NB + stop) * (R_NB / 2) * 288);
                vstop = (e + (four(za, ra, 0) + four(za, ra, 4))) + (four(zb, rb, 0) + four(zb, rb, 4));
            }
            if (__ballot(fabs(vstop) < a.tol) != 0ull) {
                uint32_t ul3 = ulane;           // (opaque copy: the chain number is made here, in the cold block, not carried through the scan)
                asm volatile("" : "+v"(ul3));
                vstop += pipe_exact_corr((uint32_t)((B0 + stop) * ((U + 1) >> 1) + (u >> 1)), a.chain0 + (uint32_t)w * 64u + ul3,
                                         a.sweep, a.seed, u & 1);
            }
            const uint32_t t = vstop > 0.0 ? 1u : 0u;
            fresh |= t << stop;
            const int sh = 8 * ((stop >> 1) & 3) + (stop & 1);
            if ((stop >> 1) < 4) rcur.x = (rcur.x & ~(1u << sh)) | (t << sh);
            else rcur.y = (rcur.y & ~(1u << sh)) | (t << sh);
            i0 = stop + 1;
        }
#pragma unroll
        for (int i = 0; i < R_NB; ++i) {
            if (i < nb) {
                const uint64_t ball = __ballot((fresh >> i) & 1u);
                if (lane == 0) a.r_bits[((int64_t)w * Nreg + B0 + i) * U + u] = ball;
                // the slot is free for block b + 2
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(const_cast<double *>(Pw) + i * 64 + ulane), R_SENT,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        rpb = spread2(fresh);
        FCD_TRACE_AT(tr0, trec, 2);
        {
            uint32_t ul2 = ulane;                 // (opaque copy: the 64-bit lane offset is made here, not carried -- spilled -- across the block)
            asm volatile("" : "+v"(ul2));
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.r_Sn + (wu * NBLK + b) * 64 + ul2);
            const unsigned long long val = (unsigned long long)rpb.x | ((unsigned long long)rpb.y << 32);
            __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // bytes, sentinels and r_bits are out before the block is announced
        if (lane == 0 && !a.withhold) __hip_atomic_store(a.flags + wu * NBLK + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        FCD_TRACE_AT(tr0, trec, 4);
        __builtin_amdgcn_s_setprio(0);
    }
}

// grid = nD + (empty workgroups beside them) + 16 nUC: workgroup (u) / (row, uc); groups of chain words one after the
// other.  The host launches it only if every workgroup is resident at once.
template <int UB, int WPE>
__global__ __launch_bounds__(1024, WPE) void gibbs_r_pipe_kernel(const r_step_args a, volatile unsigned *err) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int blk = blockIdx.x;
    if (blk < a.nD) {
        // (dsplit: two workgroups per patient, workgroup blk and blk + U: 8 chain words each)
        const int half = a.dsplit ? blk / a.u_n : -1;
        for (int wg = 0; wg < a.nWG; ++wg) pipe_diag(a, a.u_lo + (a.dsplit ? blk % a.u_n : blk), wg, half, smem, err);
    } else {
        int item = blk - a.nD;
        if (a.npad) {
            if (blk >= a.ncu + a.nD) {
                item -= a.npad;
            } else if (blk >= a.ncu) {
                return;
            }
        }
        const int row = item % R_NB, uc = item / R_NB;
        for (int wg = 0; wg < a.nWG; ++wg) pipe_panel<UB>(a, row, uc, wg, smem, err);
    }
}

// step-per-launch form: launch s = D(s-1) workgroups, then P(s) workgroups.
// Workgroups i and i + (number of CUs) land on the same CU (measured: profiles/trace_r.py), and a panel workgroup
// beside a D workgroup takes 27 us instead of 21 (beside another panel workgroup) or 16 (alone) -- it would set the
// length of the launch.  So the grid carries nD empty workgroups at [ncu, ncu + nD): the D workgroups keep their CUs
// to themselves.  (Placement is the dispatcher's business: this is a heuristic, nothing depends on it.  Letting the
// empty workgroups pack the next step's f words instead of pack_f_kernel was tried: 414 us against 400 us per pass.)
template <int UB, int WPE>
__global__ __launch_bounds__(1024, WPE) void gibbs_r_step_kernel(const r_step_args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int blk = blockIdx.x;
    if (blk < a.nD) {
        r_role_diag(a, a.s - 1, a.u_lo + blk % a.u_n, blk / a.u_n, smem);
    } else {
        int item = blk - a.nD;
        if (a.npad) {
            if (blk >= a.ncu + a.nD) {
                item -= a.npad;
            } else if (blk >= a.ncu) {
                return;
            }
        }
        const int rows = (a.Nreg - a.s * R_NB < R_NB) ? (a.Nreg - a.s * R_NB) : R_NB;
        const int nUC = (a.u_n + UB - 1) / UB;
        r_role_panel<UB>(a, a.s, item % rows, (item / rows) % nUC, item / (rows * nUC), smem);
    }
}

// ---------------------------------------------------------------------------------------------
// generic fallback (no region-major table, or a row that does not fit the LDS): one block per
// (patient, chain word), regions strictly in order, the four waves split the sum over m, direct gathers.
// ---------------------------------------------------------------------------------------------
constexpr int R_WAVES = 4;
__global__ __launch_bounds__(64 * R_WAVES) void gibbs_r_simple(const double *__restrict__ lM, const double *__restrict__ hyper,
                                                               const uint8_t *__restrict__ f_state,
                                                               uint64_t *__restrict__ r_bits, int Nreg, int U, int64_t C,
                                                               uint32_t chain0, uint64_t seed, uint32_t sweep, int mode) {
    extern __shared__ uint64_t sh_r[];
    uint64_t *mask = sh_r;                                         // [Nreg]
    double *part = reinterpret_cast<double *>(sh_r + Nreg);        // [R_WAVES][2][64]
    const int u = blockIdx.x, w = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t *__restrict__ rcol = r_bits + (int64_t)w * Nreg * U + u;
    for (int n = tid; n < Nreg; n += 64 * R_WAVES) mask[n] = rcol[(int64_t)n * U];
    __syncthreads();
    const uint8_t *__restrict__ fw = f_state + (int64_t)w * C * 64 + lane;
    const double *__restrict__ lMu = lM + (int64_t)u * 9;
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    fcd_u4 rnd = {0, 0, 0, 0};

    for (int n = 0; n < Nreg; ++n) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
        for (int m = wave; m < Nreg; m += R_WAVES) {
            const bool valid = (m != n);
            const int64_t c = valid ? fcd_pair_to_edge(n, m, mode) : 0;
            const int k = fw[c * 64];
            const uint32_t bit = (uint32_t)((mask[m] >> lane) & 1ull);
            const double *p = lMu + (c * U) * 9 + k * 3;
            const double v0 = p[bit * 2];      // r_m = 0: lM[k,0];  r_m = 1: lM[k,2]      fit.py:188-190
            const double v1 = p[2 - bit];      // r_m = 0: lM[k,2];  r_m = 1: lM[k,1]      fit.py:192-194
            s0 += valid ? v0 : 0.0;
            s1 += valid ? v1 : 0.0;
        }
        part[(wave * 2 + 0) * 64 + lane] = s0;
        part[(wave * 2 + 1) * 64 + lane] = s1;
        __syncthreads();
        if (wave == 0) {
            double t0 = part[0 * 64 + lane], t1 = part[1 * 64 + lane];
#pragma unroll
            for (int j = 1; j < R_WAVES; ++j) {
                t0 += part[(j * 2 + 0) * 64 + lane];
                t1 += part[(j * 2 + 1) * 64 + lane];
            }
            rnd = fcd_philox((uint32_t)(n * ((U + 1) >> 1) + (u >> 1)), chain, sweep, FCD_KIND_R, k0, k1);
            const double x = (u & 1) ? fcd_u53(rnd.z, rnd.w) : fcd_u53(rnd.x, rnd.y);
            const uint64_t ball = __ballot(fcd_draw_r(lnpi0 + t0, lnpi1 + t1, x));
            if (lane == 0) mask[n] = ball;
        }
        __syncthreads();
    }
    for (int n = tid; n < Nreg; n += 64 * R_WAVES) rcol[(int64_t)n * U] = mask[n];
}

template <int UB, int WPE>
int launch_step(fcd_ctx *ctx, const r_step_args &a, size_t shmem, hipStream_t s, bool prof) {
    {
        static int lds0 = 0;
        int rc0 = fcd_static_lds_check(ctx, reinterpret_cast<const void *>(&gibbs_r_step_kernel<UB, WPE>), &lds0);
        if (rc0) return rc0;
        int rc = fcd_lds_attr(ctx, FCD_KA_R_STEP + (UB == 4 ? 2 : UB - 1), reinterpret_cast<const void *>(&gibbs_r_step_kernel<UB, WPE>), shmem);
        if (rc) return rc;
    }
    if (prof) fcd_prof_begin(ctx, FCD_PROF_RSTEP, s);
    hipLaunchKernelGGL((gibbs_r_step_kernel<UB, WPE>), dim3((unsigned)(a.nD + a.nP + a.npad)), dim3(64 * a.wpb), shmem, s, a);
    if (prof) fcd_prof_end(ctx, FCD_PROF_RSTEP, s);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

// pipelined one-launch form: *fits = every workgroup resident at once (the occupancy is asked for once per shape)
template <int UB, int WPE>
int launch_pipe(fcd_ctx *ctx, const r_step_args &a, size_t shmem, bool *fits, bool launch, hipStream_t s) {
    const void *fn = reinterpret_cast<const void *>(&gibbs_r_pipe_kernel<UB, WPE>);
    const int slot = UB == 4 ? 2 : UB - 1;
    {
        static int lds0 = 0;
        int rc0 = fcd_static_lds_check(ctx, fn, &lds0);
        if (rc0) return rc0;
        int rc = fcd_lds_attr(ctx, FCD_KA_R_PIPE + slot, fn, shmem);
        if (rc) return rc;
    }
    const int threads = 64 * a.wpb;
    if (ctx->pipe_occ[slot] < 0 || ctx->pipe_occ_shmem[slot] != shmem || ctx->pipe_occ_threads[slot] != threads) {
        int per_cu = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, shmem);
        if (e != hipSuccess) return (int)e;
        ctx->pipe_occ[slot] = per_cu;
        ctx->pipe_occ_shmem[slot] = shmem;
        ctx->pipe_occ_threads[slot] = threads;
    }
    // every workgroup resident at once, with a few slots to spare (the occupancy query knows nothing of other kernels
    // on the device; a workgroup that starts late only makes the others wait -- every poll is bounded)
    *fits = (int64_t)ctx->pipe_occ[slot] * ctx->num_cu >= (int64_t)a.nD + a.nP + R_PIPE_SLOT_MARGIN;
    if (!*fits || !launch) return FCD_OK;
    fcd_prof_begin(ctx, FCD_PROF_RSTEP, s);
    if (ctx->knobs.r_coop == 1) {
        // a COOPERATIVE launch (knob r_coop = 1): the runtime itself refuses a grid that cannot be resident at once (the
        // precondition of every device-side wait in the kernel) instead of this file's occupancy arithmetic being the only
        // guard.  Not the default: measured 257.2 us per pass against 241.2 us with the plain launch
        // (profiles/r04_coop_and_collective.txt) -- the runtime serialises a cooperative dispatch against its queue.
        r_step_args ac = a;
        volatile unsigned *errp = ctx->dev_err;
        void *kargs[2] = {(void *)&ac, (void *)&errp};
        hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)(a.nD + a.nP + a.npad)), dim3(threads), kargs, (unsigned)shmem, s);
        if (e == hipErrorCooperativeLaunchTooLarge) {
            (void)hipGetLastError();
            *fits = false;                       // the caller falls back to one launch per block step
            return FCD_OK;
        }
        if (e != hipSuccess) return (int)e;
    } else {
        hipLaunchKernelGGL((gibbs_r_pipe_kernel<UB, WPE>), dim3((unsigned)(a.nD + a.nP + a.npad)), dim3(threads), shmem, s, a, ctx->dev_err);
    }
    fcd_prof_end(ctx, FCD_PROF_RSTEP, s);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

// workspace of the blocked path: P[2] | f_S | r_S | r_Sn | flags   (one formula for reserve and launch)
struct r_ws_layout {
    size_t t_bytes, f_bytes, s_bytes, flag_words, total;
};
static r_ws_layout r_ws_blocked(int64_t Nreg, int64_t U, int64_t GW) {
    r_ws_layout L;
    const int64_t NBLK = (Nreg + R_NB - 1) / R_NB;
    L.t_bytes = (size_t)GW * U * R_NB * 64 * sizeof(double);        // one buffer of panel values
    L.f_bytes = (size_t)GW * Nreg * NBLK * 64 * sizeof(uint2);
    L.s_bytes = (size_t)GW * U * NBLK * 64 * sizeof(uint2);
    L.flag_words = (size_t)GW * U * NBLK;                           // pipelined form: one mark per (word, patient, block)
    L.total = 2 * L.t_bytes + L.f_bytes + 2 * L.s_bytes + L.flag_words * sizeof(uint32_t) + 512;
    return L;
}
}  // namespace

size_t fcd_r_pass_ws_bytes(int64_t Nreg, int64_t U, int64_t GW, int r_path) {
    const size_t per_u_need = (size_t)((Nreg + R_NB - 1) / R_NB) * ((R_NB / 2) * 36 + R_NB * 6) * sizeof(double);
    if (per_u_need > 156 * 1024 || Nreg + U > 65535) return 0;          // generic kernel: no scratch
    return r_ws_blocked(Nreg, U, GW).total;
}

extern "C" int fcd_gibbs_region_tables(fcd_ctx *ctx, const double *lM, int64_t Nreg, int64_t U, int edge_mode, double *lMd,
                                       fcd_stream stream) {
    if (!ctx || !lM || !lMd) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_region_tables: null pointer");
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_region_tables: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    const int64_t total = U * Nreg * Nreg;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(region_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lM, (int)Nreg, (int)U,
                       edge_mode, lMd);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_r_step(fcd_ctx *ctx, const double *lM, const double *lMd, const double *hyper,
                                const uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                                int64_t chain0, uint64_t seed, int64_t sweep, int edge_mode, fcd_stream stream) {
    return fcd_gibbs_r_step_sq(ctx, lM, lMd, hyper, f_state, r_bits, Nreg, U, G, chain0, seed, sweep, edge_mode,
                               (hipStream_t)stream, nullptr);
}

int fcd_gibbs_r_step_sq(fcd_ctx *ctx, const double *lM, const double *lMd, const double *hyper,
                        const uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0,
                        uint64_t seed, int64_t sweep, int edge_mode, hipStream_t stream, const uint8_t *fsq,
                        const fcd_tally_f *tally_f, bool *tally_f_done, bool sentinels_in_place) {
    if (tally_f_done) *tally_f_done = false;
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, chain0, g);
    if (rc) return rc;
    if (!lM || !hyper || !f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_r_step: null pointer");
    if (ctx->dev_err && *ctx->dev_err) return fcd_fail(ctx, FCD_ERR_DEVICE, "r pass: a device-side wait was abandoned in an earlier call");
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_r_step: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    hipStream_t s = (hipStream_t)stream;
    const size_t per_u_need = (size_t)((Nreg + R_NB - 1) / R_NB) * ((R_NB / 2) * 36 + R_NB * 6) * sizeof(double);
    if (!lMd || per_u_need > 156 * 1024 || Nreg + U > 65535) {
        // generic path: direct gathers from the edge-major table
        const size_t shmem = (size_t)Nreg * 8 + (size_t)R_WAVES * 2 * 64 * 8;
        if (shmem > 64 * 1024) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: Nreg=%lld exceeds the LDS mask array", Nreg);
        if (U > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: U=%lld exceeds the grid", U);
        hipLaunchKernelGGL(gibbs_r_simple, dim3((unsigned)U, (unsigned)g.GW), dim3(64 * R_WAVES), shmem, s, lM, hyper, f_state,
                           r_bits, (int)Nreg, (int)U, g.C, (uint32_t)chain0, seed, (uint32_t)sweep, edge_mode);
        FCD_LAUNCH_CHECK();
        ctx->r_form_last = 0;
        return FCD_OK;
    }
    // blocked path.  Workspace: P[2] | f_S | r_S | r_Sn | marks
    const int NBLK = (int)((Nreg + R_NB - 1) / R_NB);
    const r_ws_layout L = r_ws_blocked(Nreg, U, g.GW);
    const size_t t_bytes = L.t_bytes, f_bytes = L.f_bytes, s_bytes = L.s_bytes;
    if ((int64_t)g.GW * Nreg * NBLK > INT32_MAX / 4 || g.C * 64 > INT32_MAX || (int64_t)g.GW * U * R_NB > INT32_MAX / 64)
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "r step: Nreg=%lld with G=%lld exceeds 32-bit item indices", Nreg, G);
    rc = fcd_ws_reserve(ctx, L.total);
    if (rc) return rc;
    char *wsp = (char *)ctx->ws;
    double *Pb[2] = {(double *)wsp, (double *)(wsp + t_bytes)};
    wsp += 2 * t_bytes;
    uint2 *f_S = (uint2 *)wsp;
    wsp += f_bytes;
    uint2 *r_S = (uint2 *)wsp, *r_Sn = (uint2 *)(wsp + s_bytes);
    wsp += 2 * s_bytes;
    uint32_t *marks = (uint32_t *)wsp;
    fcd_abl_refresh(s);
    // patients per panel workgroup: the pair tile (288 B per pair of regions) + the single rows must fit the LDS
    const size_t per_u = (size_t)NBLK * ((R_NB / 2) * 36 + R_NB * 6) * sizeof(double);
    // Two workgroups must fit a CU so that their staging / pair-build / term phases overlap: 2 patients where their
    // tile takes at most half the LDS (cfg3: 2 x 39.9 KB), else 1 (cfg5: 76.8 KB per patient; measured 3.99 ms per pass
    // against 4.58 ms with 2 patients and one workgroup per CU)
    int ub = ((size_t)4 * per_u <= 160 * 1024 && U >= 2) ? 2 : 1;
    {   // tuning knob: patients per panel workgroup (1, 2, 4)
        const int v = ctx->knobs.r_ub;
        if ((v == 1 || v == 2 || v == 4) && (size_t)v * per_u <= 156 * 1024) ub = v;
    }
    size_t shmem = (size_t)ub * per_u;
    {
        const size_t d_need = (size_t)(g.GW < 16 ? D_LDS_SPREAD : D_LDS_COMPACT) * sizeof(double);
        if (shmem < d_need) shmem = d_need;
    }
    r_step_args a;
    a.lMd = lMd; a.hyper = hyper; a.f_S = f_S;
    a.r_S = r_S; a.r_Sn = r_Sn; a.r_bits = r_bits;
    a.Pbuf[0] = Pb[0]; a.Pbuf[1] = Pb[1];
    a.flags = nullptr;
    a.Nreg = (int)Nreg; a.U = (int)U; a.NBLK = NBLK; a.GW = g.GW;
    a.u_lo = 0; a.u_n = (int)U;
    a.wpb = g.GW < 16 ? g.GW : 16;
    a.nWG = (g.GW + a.wpb - 1) / a.wpb;
    a.s = 0; a.nD = 0; a.nP = 0;
    a.ncu = ctx->num_cu; a.npad = 0;
    a.chain0 = (uint32_t)chain0; a.sweep = (uint32_t)sweep; a.seed = seed;
    a.tol = 16.0 * FCD_LOGIT_FAST_ERR;
    a.poll_limit = ctx->knobs.r_poll_limit > 0 ? ctx->knobs.r_poll_limit : R_POLL_LIMIT;
    a.withhold = ctx->knobs.r_withhold;
    if (ctx->knobs.r_tol > a.tol) a.tol = ctx->knobs.r_tol;   // test hook: a huge value sends every draw through the exact path
    const int nUC = (int)((U + ub - 1) / ub);
    // Pipelined one-launch form (the default where it fits; knob r_path = 3 keeps the step-per-launch form): needs every
    // workgroup resident at once (with a few slots to spare) and the pinned error word.
    bool pipe = false;
    r_pipe_init pinit;
    pinit.marks = nullptr; pinit.P[0] = pinit.P[1] = nullptr;
    a.dsplit = 0;
    if (ctx->knobs.r_path != 3 && ctx->dev_err) {
        a.nD = (int)U;
        a.nP = R_NB * nUC;
        a.npad = (!ctx->knobs.r_nopad && a.nD <= a.ncu && a.nD + a.nP > a.ncu) ? a.nD : 0;
        // The in-order role is the serial chain of the pass and bound by the vector instructions ONE CU issues for a
        // patient's 16 chain words: with more than 8 words per group give every patient two workgroups (8 words each,
        // on two CUs beside a panel workgroup each, instead of one workgroup beside an empty one) -- knob r_dsplit = 1 keeps one.
        if (ctx->knobs.r_dsplit != 1 && a.wpb > 8 && 2 * a.nD <= a.ncu) {
            a.dsplit = 1;
            a.nD = 2 * (int)U;
            a.npad = 0;
        }
        if (ub == 4) rc = launch_pipe<4, 4>(ctx, a, shmem, &pipe, false, s);
        else if (ub == 2) rc = launch_pipe<2, 8>(ctx, a, shmem, &pipe, false, s);
        else rc = launch_pipe<1, 8>(ctx, a, shmem, &pipe, false, s);
        if (rc) return rc;
        if (pipe) {
            pinit.marks = marks;
            // (the sentinels: unless a completed pipelined pass of this shape left them in place -- fcd_gibbs_run knows)
            const bool keep = sentinels_in_place && ctx->r_form_last == 2 && !ctx->knobs.r_refill;
            pinit.P[0] = keep ? nullptr : Pb[0];
            pinit.P[1] = keep ? nullptr : Pb[1];
        }
    }
    {
        // one launch packs the f words of every region and the r words of every patient
        const int fb = fsq ? 2 : 1;                 // blocks per wave (pack_f_kernel)
        dim3 pgrid((unsigned)(((NBLK + fb - 1) / fb + 3) / 4), (unsigned)(Nreg + U), (unsigned)g.GW);
        // ... and, asked to, the f half of the sweep's tally in extra rows of workgroups (about two per CU: four waves each,
        // four edges per wave and round)
        fcd_tally_f tf;
        tf.f_state = nullptr; tf.C = 0; tf.G = 0; tf.GW = 0; tf.acc = nullptr; tf.cnt_f = nullptr;
        // (only where the f state is small beside the packing's own work: at cfg5 -- 79 800 edges x 16 chain words -- the extra
        // rows cost the launch 66 us for 16 us saved in the tally after the pass: profiles/r03_kernel_stats_cfg5.txt)
        if (tally_f && g.C * g.GW <= 600000) {
            tf = *tally_f;
            const int64_t per_row = (int64_t)pgrid.x * pgrid.z;
            int64_t want = (g.C + 15) / 16;                       // workgroups of one round
            if (want > 2 * (int64_t)ctx->num_cu) want = 2 * (int64_t)ctx->num_cu;
            int64_t ty = (want + per_row - 1) / per_row;
            if (ty < 1) ty = 1;
            if (pgrid.y + ty <= 65535) {
                pgrid.y += (unsigned)ty;
                if (tally_f_done) *tally_f_done = true;
            } else {
                tf.f_state = nullptr;
            }
        }
        fcd_prof_begin(ctx, FCD_PROF_PACK, s);
        if (fsq)
            hipLaunchKernelGGL(pack_f_kernel<true>, pgrid, dim3(256), 0, s, fsq, (int)Nreg, NBLK, (int)g.C, edge_mode, f_S, r_bits,
                               (int)U, r_S, pinit, tf);
        else
            hipLaunchKernelGGL(pack_f_kernel<false>, pgrid, dim3(256), 0, s, f_state, (int)Nreg, NBLK, (int)g.C, edge_mode, f_S,
                               r_bits, (int)U, r_S, pinit, tf);
        fcd_prof_end(ctx, FCD_PROF_PACK, s);
        FCD_LAUNCH_CHECK();
    }
    ctx->r_form_last = pipe ? 2 : 1;
    if (pipe) {
        a.flags = pinit.marks;
        if (ub == 4) rc = launch_pipe<4, 4>(ctx, a, shmem, &pipe, true, s);
        else if (ub == 2) rc = launch_pipe<2, 8>(ctx, a, shmem, &pipe, true, s);
        else rc = launch_pipe<1, 8>(ctx, a, shmem, &pipe, true, s);
        if (rc || pipe) return rc;
        // the runtime refused the cooperative launch (grid not co-resident after all): the step-per-launch form instead
        ctx->r_form_last = 1;
        a.flags = nullptr;
        a.dsplit = 0;
    }
    // one launch per block step: launch st = D(st-1) workgroups + P(st) workgroups
    for (int st = 0; st <= NBLK; ++st) {
        const int rows = st < NBLK ? (int)((Nreg - (int64_t)st * R_NB < R_NB) ? (Nreg - (int64_t)st * R_NB) : R_NB) : 0;
        a.s = st;
        a.nD = st >= 1 ? (int)U * a.nWG : 0;
        a.nP = rows * nUC * a.nWG;
        a.npad = (!ctx->knobs.r_nopad && a.nD > 0 && a.nD <= a.ncu && a.nD + a.nP > a.ncu) ? a.nD : 0;
        if (a.nD + a.nP == 0) continue;
        if (ub == 4) rc = launch_step<4, 4>(ctx, a, shmem, s, true);
        else if (ub == 2) rc = launch_step<2, 8>(ctx, a, shmem, s, true);
        else rc = launch_step<1, 8>(ctx, a, shmem, s, true);
        if (rc) return rc;
    }
    return FCD_OK;
}
