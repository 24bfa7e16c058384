// Many-chain collapsed Gibbs sampler over (f, r) for the IAR model, built on the reference's per-edge
// tables.  The two conditionals are fcdiff/fit.py:170-173 (f) and :187-194 (r) at one-hot q.
//
// Mapping: lane = chain.  A wave owns one "chain word" (64 chains); everything that does not depend
// on the chain (edge id, patient, region pair, table tile) is wave-uniform and lives in SGPRs / LDS:
//   * r is stored as bit planes r_bits[w][n][u] (one uint64 = r_nu of the 64 chains of word w), so
//     the mixture index l(r_nu, r_mu) of all 64 chains comes from two scalar masks and becomes a
//     per-lane LDS byte offset with two v_cndmask;
//   * f is stored as f_state[w][c][64] bytes: a wave reads/writes 64 contiguous bytes per edge.
//
// Algorithmic bytes per sweep of G chains: 2*72*C*U (lM once per pass) + 24*C +
// G*(2*C + 3*Nreg*U) of state.  Arithmetic: 3 (f) + 4 (r) fp64 adds per (edge, patient, chain).
#include "fcd_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// init
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gibbs_init_f(uint8_t *__restrict__ f_state, int64_t C, int GW, uint32_t chain0,
                                                    uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (w, pair of edges)
    const int64_t n_pairs = (C + 1) / 2;
    if (item >= n_pairs * GW) return;
    const int w = (int)(item / n_pairs);
    const int64_t pr = item % n_pairs;
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const fcd_u4 x = fcd_philox((uint32_t)pr, chain, 0u, FCD_KIND_INIT_F, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int f0 = min((int)(fcd_u53(x.x, x.y) * 3.0), 2);
    const int f1 = min((int)(fcd_u53(x.z, x.w) * 3.0), 2);
    const int64_t c = pr * 2;
    f_state[((int64_t)w * C + c) * 64 + lane] = (uint8_t)f0;
    if (c + 1 < C) f_state[((int64_t)w * C + c + 1) * 64 + lane] = (uint8_t)f1;
}

__global__ __launch_bounds__(256) void gibbs_init_r(uint64_t *__restrict__ r_bits, int Nreg, int U, int GW,
                                                    uint32_t chain0, uint64_t seed, double pi) {
    const int lane = threadIdx.x & 63;
    const int64_t n_np = (Nreg + 1) / 2;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (w, region pair, u)
    if (item >= n_np * U * GW) return;
    const int w = (int)(item / (n_np * U));
    const int64_t rest = item % (n_np * U);
    const int np = (int)(rest / U), u = (int)(rest % U);
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const fcd_u4 x = fcd_philox((uint32_t)(np * U + u), chain, 0u, FCD_KIND_INIT_R, (uint32_t)seed,
                                (uint32_t)(seed >> 32));
    const uint64_t b0 = __ballot(fcd_u53(x.x, x.y) < pi);
    const uint64_t b1 = __ballot(fcd_u53(x.z, x.w) < pi);
    if (lane == 0) {
        const int n = np * 2;
        r_bits[((int64_t)w * Nreg + n) * U + u] = b0;
        if (n + 1 < Nreg) r_bits[((int64_t)w * Nreg + n + 1) * U + u] = b1;
    }
}

// ---------------------------------------------------------------------------------------------
// f step.  grid.x = edge tiles of Ec edges, grid.y = groups of (blockDim/64) chain words.
// The tile lM[c0 : c0+Ec, :, :, :] (contiguous, Ec*U*72 bytes) is staged in LDS once and read by every
// wave of the block: with 16 waves (1024 chains) per block the table is read from HBM/L2 once per pass.
// COND = true writes the three unnormalised log-weights instead of drawing (parity hook).
// ---------------------------------------------------------------------------------------------
constexpr int F_UNROLL = 8;
template <bool COND>
__global__ __launch_bounds__(1024) void gibbs_f_kernel(const double *__restrict__ S_B, const double *__restrict__ lM,
                                                       const double *__restrict__ hyper, uint8_t *__restrict__ f_state,
                                                       const uint64_t *__restrict__ r_bits, int Nreg, int U, int64_t C,
                                                       int GW, int64_t G, int Ec, uint32_t chain0, uint64_t seed,
                                                       uint32_t sweep, double *__restrict__ cond_f) {
    extern __shared__ double tile[];
    const int64_t c0 = (int64_t)blockIdx.x * Ec;
    const int ne = (int)((C - c0 < Ec) ? (C - c0) : Ec);
    {
        const int64_t n_dbl = (int64_t)ne * U * 9;
        const double *src = lM + c0 * U * 9;
        for (int64_t i = threadIdx.x; i < n_dbl; i += blockDim.x) tile[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= GW) return;
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const double lng0 = hyper[FCD_H_LNGAMMA + 0], lng1 = hyper[FCD_H_LNGAMMA + 1], lng2 = hyper[FCD_H_LNGAMMA + 2];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint64_t *__restrict__ rw = r_bits + (int64_t)w * Nreg * U;
    fcd_u4 rnd = {0, 0, 0, 0};
    int64_t rnd_idx = -1;

    for (int e = 0; e < ne; ++e) {
        const int64_t c = c0 + e;
        int n, m;
        fcd_edge_to_pair(c, n, m);
        const uint64_t *__restrict__ rn = rw + (int64_t)n * U;
        const uint64_t *__restrict__ rm = rw + (int64_t)m * U;
        const char *tb = reinterpret_cast<const char *>(tile + (int64_t)e * U * 9);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        // l = 1 (both anomalous) -> +8 B, l = 2 (discordant) -> +16 B, l = 0 -> +0 inside lM[c,u,k,:].
        // Chunks of F_UNROLL patients with a fixed trip count: the masks of a chunk arrive in one wide
        // scalar load per region and the LDS reads of a chunk are all in flight together.
        int u = 0;
        for (; u + F_UNROLL <= U; u += F_UNROLL) {
#pragma unroll
            for (int j = 0; j < F_UNROLL; ++j) {
                const uint64_t mn = rn[u + j], mm = rm[u + j];
                const uint32_t off = fcd_sel_mask(fcd_sel_mask(0u, 16u, mn ^ mm), 8u, mn & mm);
                const double *p = reinterpret_cast<const double *>(tb + (u + j) * 72 + off);
                a0 += p[0];
                a1 += p[3];
                a2 += p[6];
            }
        }
        for (; u < U; ++u) {
            const uint64_t mn = rn[u], mm = rm[u];
            const uint32_t off = fcd_sel_mask(fcd_sel_mask(0u, 16u, mn ^ mm), 8u, mn & mm);
            const double *p = reinterpret_cast<const double *>(tb + u * 72 + off);
            a0 += p[0];
            a1 += p[3];
            a2 += p[6];
        }
        a0 = lng0 + (S_B[c * 3 + 0] + a0);   // fit.py:165, 171-173
        a1 = lng1 + (S_B[c * 3 + 1] + a1);
        a2 = lng2 + (S_B[c * 3 + 2] + a2);
        if (COND) {
            if ((int64_t)w * 64 + lane < G) {
                double *o = cond_f + (((int64_t)w * 64 + lane) * C + c) * 3;
                o[0] = a0; o[1] = a1; o[2] = a2;
            }
        } else {
            if ((c >> 2) != rnd_idx) {
                rnd_idx = c >> 2;
                rnd = fcd_philox((uint32_t)rnd_idx, chain, sweep, FCD_KIND_F, k0, k1);
            }
            const double x = fcd_u32(fcd_word(rnd, (int)(c & 3)));
            f_state[((int64_t)w * C + c) * 64 + lane] = (uint8_t)fcd_draw_f(a0, a1, a2, x);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// f step, difference form.  Only the two log-odds b_k = a_k - a_0, k = 1, 2 enter the draw, and
//   b_k = ln(gamma_k/gamma_0) + (S_B[c,k] - S_B[c,0]) + sum_u lMf[c,u,l_u,k-1],
//   lMf[c,u,l,k-1] = lM[c,u,k,l] - lM[c,u,0,l]            (edge-major difference table, 48 B per (c,u))
// so a term is one 16-byte LDS read and two fp64 adds.  Same tiling as gibbs_f_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_tables_kernel(const double *__restrict__ lM, int64_t n_items,
                                                          double *__restrict__ lMf) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        const double *p = lM + i * 9;
        double *o = lMf + i * 6;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            o[l * 2 + 0] = p[3 + l] - p[l];
            o[l * 2 + 1] = p[6 + l] - p[l];
        }
    }
}

__global__ __launch_bounds__(1024) void gibbs_f_diff_kernel(const double *__restrict__ S_B, const double *__restrict__ lMf,
                                                            const double *__restrict__ hyper, uint8_t *__restrict__ f_state,
                                                            const uint64_t *__restrict__ r_bits, int Nreg, int U, int64_t C,
                                                            int GW, int Ec, uint32_t chain0, uint64_t seed, uint32_t sweep,
                                                            float margin) {
    extern __shared__ __attribute__((aligned(16))) double tile[];   // [Ec][U][3][2]
    const int64_t c0 = (int64_t)blockIdx.x * Ec;
    const int ne = (int)((C - c0 < Ec) ? (C - c0) : Ec);
    {
        const int n_d2 = ne * U * 3;
        const double2 *src = reinterpret_cast<const double2 *>(lMf + c0 * U * 6);
        double2 *dst = reinterpret_cast<double2 *>(tile);
        for (int i = threadIdx.x; i < n_d2; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= GW) return;
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const double lg1 = hyper[FCD_H_LNGAMMA + 1] - hyper[FCD_H_LNGAMMA + 0];
    const double lg2 = hyper[FCD_H_LNGAMMA + 2] - hyper[FCD_H_LNGAMMA + 0];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint64_t *__restrict__ rw = r_bits + (int64_t)w * Nreg * U;
    fcd_u4 rnd = {0, 0, 0, 0};
    int64_t rnd_idx = -1;
    if (FCD_ABL(0, 3)) return;           // ablation: staging only

    for (int e = 0; e < ne; ++e) {
        const int64_t c = c0 + e;
        int n, m;
        fcd_edge_to_pair(c, n, m);
        const uint64_t *__restrict__ rn = rw + (int64_t)n * U;
        const uint64_t *__restrict__ rm = rw + (int64_t)m * U;
        const char *tb = reinterpret_cast<const char *>(tile) + (int64_t)e * U * 48;
        double b1 = 0.0, b2 = 0.0;
        int u = 0;
        for (; u + F_UNROLL <= U; u += F_UNROLL) {
#pragma unroll
            for (int j = 0; j < F_UNROLL; ++j) {
                const uint64_t mn = rn[u + j], mm = rm[u + j];
                // l = 1 (both anomalous) -> +16 B, l = 2 (discordant) -> +32 B, l = 0 -> +0 inside lMf[c,u,:,:]
                const uint32_t off = fcd_sel_mask(fcd_sel_mask(0u, 32u, mn ^ mm), 16u, mn & mm);
                const double2 v = *reinterpret_cast<const double2 *>(tb + (u + j) * 48 + off);
                b1 += v.x;
                b2 += v.y;
            }
        }
        for (; u < U; ++u) {
            const uint64_t mn = rn[u], mm = rm[u];
            const uint32_t off = fcd_sel_mask(fcd_sel_mask(0u, 32u, mn ^ mm), 16u, mn & mm);
            const double2 v = *reinterpret_cast<const double2 *>(tb + u * 48 + off);
            b1 += v.x;
            b2 += v.y;
        }
        b1 = lg1 + ((S_B[c * 3 + 1] - S_B[c * 3 + 0]) + b1);
        b2 = lg2 + ((S_B[c * 3 + 2] - S_B[c * 3 + 0]) + b2);
        if (FCD_ABL(0, 2)) {             // ablation: no RNG / exp
            f_state[((int64_t)w * C + c) * 64 + lane] = (uint8_t)(b1 > b2 ? 1 : 2);
            continue;
        }
        if ((c >> 2) != rnd_idx) {
            rnd_idx = c >> 2;
            rnd = fcd_philox((uint32_t)rnd_idx, chain, sweep, FCD_KIND_F, k0, k1);
        }
        const double x = fcd_u32(fcd_word(rnd, (int)(c & 3)));
        bool amb;
        int k = fcd_draw_f_fast(b1, b2, x, 2e-5f, margin, &amb);
        if (__ballot(amb) != 0ull) k = fcd_draw_f(0.0, b1, b2, x);      // too close to a boundary somewhere in the wave
        f_state[((int64_t)w * C + c) * 64 + lane] = (uint8_t)k;
    }
}

// ---------------------------------------------------------------------------------------------
// f step, pair form (U <= 64).  The kernel is bound by wave-wide LDS reads (one per gathered value whatever
// the number of distinct addresses: profiles/r01_ubench_lds_fp64.txt), so the tile holds records for PAIRS of
// patients (u, u+1): for each of the 16 values of (x_u, x_u+1, a_u, a_u+1), x = r_n xor r_m, a = r_n and r_m,
//   rec[slot] = lMf[c][u][l(x_u, a_u)][:] + lMf[c][u+1][l(x_u+1, a_u+1)][:]        (2 doubles, 256 B per pair)
// built in LDS while staging.  One ds_read_b128 + two fp64 adds then cover two patients.  r comes as per-lane
// words over patients (r_U, made by pack_ru_kernel): no scalar loads inside the loop.  A word holds 16 patients
// SPREAD over 4-bit fields, the pair (u, u+1) in the low two bits of field u/2: (r_n ^ r_m) | (r_n & r_m) << 2 is
// then the slot number of every pair at once, and a term costs two integer instructions for its address.
// ---------------------------------------------------------------------------------------------
// One slot-source word of (w, n): 16 patients in 4-bit fields (the pair (u, u+1) in the low two bits of field u/2).
// Where slot word jw of (w, n) = wn lives for lane: the words of a region are kept FOUR side by side per lane (round 4), so
// that one 16-byte load fetches what four 4-byte loads did -- a vector-memory instruction occupies the CU's address unit
// for the same ~16 cycles whatever its width (profiles/r04_ubench_vmem_rate.txt), and the pair kernel issued eight per edge.
// (NW <= 4, the U <= 64 kernel: side by side.  The any-U kernel consumes one word per loop turn and keeps [word][lane]: with
// the 16-byte words held across four turns it measured 1.83 against 1.72 ms at cfg5.)
__host__ __device__ static inline int64_t ru_index(int64_t wn, int NW, int jw, int lane) {
    return NW <= 4 ? (wn * 64 + lane) * 4 + jw : (wn * NW + jw) * 64 + lane;
}
__device__ __forceinline__ uint32_t pack_ru_word(const uint64_t *__restrict__ r_bits, int64_t wn, int U, int jw, int lane) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int u = jw * 16 + j;
        const uint64_t word = r_bits[wn * U + (u < U ? u : U - 1)];                              // clamped: no branch per load
        v |= (u < U ? (uint32_t)((word >> lane) & 1ull) : 0u) << (4 * (j >> 1) + (j & 1));
    }
    return v;
}
__global__ __launch_bounds__(256) void pack_ru_kernel(const uint64_t *__restrict__ r_bits, int Nreg, int U, int NW, int GW,
                                                      uint32_t *__restrict__ r_U) {
    const int lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));      // (w, n, word): scalar
    if (item >= GW * Nreg * NW) return;
    const int jw = item % NW, wn = item / NW;                   // wn = w*Nreg + n
    r_U[ru_index(wn, NW, jw, lane)] = pack_ru_word(r_bits, wn, U, jw, lane);
}

constexpr int FP_EC = 8;     // edges per tile
typedef float fcd_f2v __attribute__((ext_vector_type(2)));           // one ds_read_b64, one v_pk_add_f32
typedef __attribute__((address_space(3))) const fcd_f2v lds_cf2v;

// The exact path of the pair forms: the two log-odds sums of edge c for this lane's chain in fp64, straight from the
// table (48 contiguous bytes per patient: L2), in the order of the fp64 pair records of rounds 2-3 (record = row u + row
// u+1, records summed in pair order).  zs: the slot word of a group of 8 pairs, np pairs of it used.
__device__ __forceinline__ void fcd_f_exact_group(const double *__restrict__ row, int U, int u0, int np, uint32_t zs,
                                                  bool &first, double &b1, double &b2) {
#pragma unroll 1
    for (int p = 0; p < np; ++p) {
        const uint32_t f = (zs >> (4 * p)) & 15u;
        const int l0 = (f & 4u) ? 1 : ((f & 1u) ? 2 : 0), l1 = (f & 8u) ? 1 : ((f & 2u) ? 2 : 0);   // 0 typical, 1 both, 2 discordant
        const int u = u0 + 2 * p;
        const double2 va = *reinterpret_cast<const double2 *>(row + ((int64_t)u * 3 + l0) * 2);
        double r1 = va.x, r2 = va.y;
        if (u + 1 < U) {
            const double2 vb = *reinterpret_cast<const double2 *>(row + ((int64_t)(u + 1) * 3 + l1) * 2);
            r1 += vb.x;
            r2 += vb.y;
        }
        if (first) {
            b1 = r1;
            b2 = r2;
            first = false;
        } else {
            b1 += r1;
            b2 += r2;
        }
    }
}

// The exact path of the U <= 64 kernel as ONE out-of-line function: inlined into the eight unrolled edge bodies it was most of
// the kernel's text (fp64 exponentials, the gather loop), and the hot path had to be fetched around it.
__device__ __attribute__((noinline)) int fcd_f_exact_edge(const double *__restrict__ row, double c1, double c2, uint32_t z0, uint32_t z1,
                                                          uint32_t z2, uint32_t z3, int U, uint32_t xw) {
    const int NPAIR = (U + 1) >> 1, NG = (NPAIR + 7) >> 3;
    bool first = true;
    double b1 = 0.0, b2 = 0.0;
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const uint32_t zs = g == 0 ? z0 : g == 1 ? z1 : g == 2 ? z2 : z3;
        fcd_f_exact_group(row, U, 16 * g, min(8, NPAIR - 8 * g), zs, first, b1, b2);
    }
    return fcd_draw_f(0.0, c1 + b1, c2 + b2, fcd_u32(xw));
}

template <int NW16>
__global__ __launch_bounds__(1024, 8) void gibbs_f_pair_kernel(const double *__restrict__ S_B, const double *__restrict__ lMf,
                                                            const double *__restrict__ hyper, uint8_t *__restrict__ f_state,
                                                            const uint32_t *__restrict__ r_U, int Nreg, int U, int64_t C,
                                                            int GW, uint32_t chain0, uint64_t seed, uint32_t sweep, float margin, uint8_t *__restrict__ fsq,
                                                            unsigned long long *__restrict__ dbg) {
    // pair records [NPAIR][FP_EC][16] float2 | per-edge constants [FP_EC] float4 | singles [FP_EC][U][3][2] double
    extern __shared__ __attribute__((aligned(256))) double ptile[];
    const int NPAIR = (U + 1) >> 1;
    const int64_t c0 = (int64_t)blockIdx.x * FP_EC;
    const int ne = (int)((C - c0 < FP_EC) ? (C - c0) : FP_EC);
    float4 *edge_k = reinterpret_cast<float4 *>(reinterpret_cast<char *>(ptile) + (size_t)FP_EC * NPAIR * 128);
    double2 *single = reinterpret_cast<double2 *>(reinterpret_cast<char *>(edge_k) + FP_EC * 16);
    const double lg1 = hyper[FCD_H_LNGAMMA + 1] - hyper[FCD_H_LNGAMMA + 0];
    const double lg2 = hyper[FCD_H_LNGAMMA + 2] - hyper[FCD_H_LNGAMMA + 0];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    [[maybe_unused]] const int trec = (int)blockIdx.x < 4096 ? (int)blockIdx.x : 4095;      // (diagnostic build: profiles/trace_f.py)
    FCD_TRACE(trec, 0);
    FCD_TRACE_VAL(trec, 7, (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff));
    // the r words of the tile's edges: loaded before the barrier so their latency hides behind the staging.
    // Consecutive edges c = n(n-1)/2 + m share n: its words are fetched once per run, not once per edge.
    // Slot numbers (x_u, x_u+1, a_u, a_u+1) of 8 pairs of patients per word, for the edge at hand only: the words of
    // the NEXT edge are requested while this edge's terms run (32 registers less than holding all 8 edges' words).
    uint32_t Zc[NW16], rn[NW16];
    int wn, wm;                          // (n, m) of the edge at hand (wave-uniform walk of the lower-triangular order)
    fcd_edge_to_pair(c0, wn, wm);
    // wave-uniform base + unsigned 32-bit (region, lane) offset: no per-lane 64-bit address arithmetic.  The (up to four)
    // slot words of a region are ONE 16-byte load (ru_index).
    const uint4 *__restrict__ ru = reinterpret_cast<const uint4 *>(r_U) + (int64_t)(w < GW ? w : 0) * Nreg * 64;
    const uint32_t ul = (uint32_t)lane;
    auto comp = [](const uint4 &v, int j) -> uint32_t { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
    {
        const uint4 vn = ru[(uint32_t)(wn * 64) + ul], vm = ru[(uint32_t)(wm * 64) + ul];
#pragma unroll
        for (int j = 0; j < NW16; ++j) {
            rn[j] = comp(vn, j);
            const uint32_t rm = comp(vm, j);
            Zc[j] = (rn[j] ^ rm) | ((rn[j] & rm) << 2);
        }
    }
    // S_B of the tile's edges (lane l < 3 of wave e: S_B[c0 + e][l]): asked for HERE, beside the rows -- the build phase turns them
    // into the edge's constants, and a trip to memory in the middle of that phase sat on every tile's critical path
    // (profiles/r04_trace_f.txt: built 3.8 us of a 17 us tile)
    const int wv0 = (int)(threadIdx.x >> 6);
    double sbv = 0.0;
    if (wv0 < ne && lane < 3) sbv = S_B[(c0 + wv0) * 3 + lane];
    {
        // the tile's rows of lMf are one contiguous piece: coalesced 16-byte copies
        const int n_d2 = ne * U * 3;
        const double2 *src = reinterpret_cast<const double2 *>(lMf + c0 * U * 6);
        // (two loads per thread in flight before the first store: a tile is at most 1536 pieces)
        for (int i0 = threadIdx.x; i0 < n_d2; i0 += 2 * blockDim.x) {
            const int i1 = i0 + (int)blockDim.x;
            const double2 v0 = src[i0], v1 = src[i1 < n_d2 ? i1 : i0];
            single[i0] = v0;
            if (i1 < n_d2) single[i1] = v1;
        }
    }
    __syncthreads();
    FCD_TRACE(trec, 1);
    {
        // pair records from the single rows in LDS, ONE THREAD PER (edge, pair): six 16-byte reads (the three mixture cases of
        // patient u and of patient u+1), the nine valid sums, rounded once to fp32, eight 16-byte writes (two slots each; the
        // seven impossible slots -- x and a both set for a patient -- as zeros).  Round 3's build gave every (pair, slot) entry a
        // thread of its own: 16 x the index arithmetic and 7/16 of the threads making zeros -- 3.7 us of a 17 us tile
        // (profiles/r04_trace_f.txt).
        typedef float fcd_f4v __attribute__((ext_vector_type(4)));
        fcd_f4v *dst4 = reinterpret_cast<fcd_f4v *>(ptile);
        const int total = ne * NPAIR;
        // (the records are made by the UPPER half of the workgroup's waves: the lower ones make the edges' bounds below)
        const int half = (int)(blockDim.x >> 7) << 6;
        const int tb = (int)threadIdx.x >= half ? (int)threadIdx.x - half : (int)threadIdx.x + ((int)blockDim.x - half);
        for (int ep = tb; ep < total; ep += blockDim.x) {
            const int e = ep / NPAIR, pr = ep - e * NPAIR;
            const int u = 2 * pr;
            const double2 *su = single + (e * U + u) * 3;
            double2 A[3], B[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                A[l] = su[l];
                B[l] = (u + 1 < U) ? su[3 + l] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                fcd_f4v o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int slot = 2 * s2 + h;
                    const int x0 = slot & 1, x1 = (slot >> 1) & 1, a0 = (slot >> 2) & 1, a1 = slot >> 3;
                    const bool valid = !((x0 & a0) | (x1 & a1));
                    const int l0 = a0 ? 1 : (x0 ? 2 : 0), l1 = a1 ? 1 : (x1 ? 2 : 0);          // 0 typical, 1 both, 2 discordant
                    const float vx = valid ? (float)(A[l0].x + B[l1].x) : 0.f, vy = valid ? (float)(A[l0].y + B[l1].y) : 0.f;
                    if (h == 0) { o.x = vx; o.y = vy; } else { o.z = vx; o.w = vy; }
                }
                dst4[(pr * FP_EC + e) * 8 + s2] = o;       // [pair][edge][slot]: see the reads below
            }
        }
        // B_e >= sum over the pairs of max |record entry|: the sum over the patients of the largest |value| of the row
        // (wave e does edge e, lane = patient; U <= 64) -- what bounds the error of the fp32 sums below
        for (int ee = (int)(threadIdx.x >> 6); ee < ne; ee += (int)(blockDim.x >> 6)) {      // (a workgroup may have fewer waves than edges)
            float a = 0.f;
            if (lane < U) {
                const double2 *su = single + (ee * U + lane) * 3;
#pragma unroll
                for (int l = 0; l < 3; ++l) a = fmaxf(a, fmaxf((float)fabs(su[l].x), (float)fabs(su[l].y)));
                a *= 1.0000002f;                              // (the conversions above round to nearest)
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0) {
                // the edge's constants as the draw wants them: the two log-odds offsets in fp32 and the relative uncertainty
                // of the weights that the error bound of the whole fp32 sum (NPAIR records, then the offset) amounts to
                const int64_t c = c0 + ee;
                double sb0, sb1, sb2;
                if (ee == wv0) {                       // (this wave's own edge: the values asked for at the top)
                    const uint64_t bits = (uint64_t)__double_as_longlong(sbv);      // lane 0 holds S_B[c][0]; lanes 1, 2 the others
                    auto lane_val = [&](int l) -> double {
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), l);
                        return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
                    };
                    sb0 = lane_val(0); sb1 = lane_val(1); sb2 = lane_val(2);
                } else {
                    sb0 = S_B[c * 3 + 0]; sb1 = S_B[c * 3 + 1]; sb2 = S_B[c * 3 + 2];
                }
                const double c1 = lg1 + (sb1 - sb0), c2 = lg2 + (sb2 - sb0);
                const float cm = fmaxf((float)fabs(c1), (float)fabs(c2)) * 1.0000002f;
                edge_k[ee] = make_float4((float)c1, (float)c2, fcd_draw_f_eta(fcd_f32_sum_err(NPAIR, a) + fcd_f32_offset_err(cm, a)), 0.f);
            }
        }
    }
    __syncthreads();
    FCD_TRACE(trec, 2);
    if (w >= GW) return;
    if (FCD_ABL(0, 3)) return;           // ablation: staging only
#ifdef FCD_ABLATE
    long long tr_terms = 0, tr_draw = 0, tr_rest = 0;
#endif
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    fcd_u4 rnd = {0, 0, 0, 0};
    const int NG = (NPAIR + 7) >> 3;     // groups of 8 pairs = 16 patients = one slot word
    // The records lie [pair][edge][slot] from LDS address 0 (the kernel declares no static LDS: the host asks the runtime
    // before the first launch), so with the loops over edges, groups and pairs unrolled the record's address is an
    // IMMEDIATE of the read and a term's address costs a shift and a mask -- whatever the number of patients.

#pragma unroll
    for (int e = 0; e < FP_EC; ++e) {
        if (e < ne) {
            const int64_t c = c0 + e;
            // next edge: (n, m+1), or (n+1, 0) at the end of row n (clamped past the last edge: unused there)
            int nn = wn, nm = wm + 1;
            if (nm == nn) {
                nm = 0;
                nn = (nn + 1 < Nreg) ? nn + 1 : nn;
            }
            // the next edge's words: region m always, region n only where the row of the triangle changes (wave-uniform)
            uint4 vmn = make_uint4(0u, 0u, 0u, 0u), vnn = vmn;
            const bool new_row = nn != wn;
            if (e + 1 < FP_EC) {
                vmn = ru[(uint32_t)(nm * 64) + ul];
                if (new_row) vnn = ru[(uint32_t)(nn * 64) + ul];
            }
#ifdef FCD_ABLATE
            const long long tc0 = clock64();
#endif
            fcd_f2v acc = {0.f, 0.f}, acc1 = {0.f, 0.f};    // even / odd pairs: two chains of packed adds, none waits for the one before
#pragma unroll
            for (int g = 0; g < NW16; ++g) {
                if (g < NG) {
                    const uint32_t zs = Zc[g];
                    if (NPAIR - 8 * g >= 8) {
                        // eight reads asked for together, then the eight packed adds
                        fcd_f2v v[8];
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            // slot -> 8-byte records: byte offset = slot << 3
                            const uint32_t ad = (p == 0) ? (zs << 3) & 0x78u : (zs >> (4 * p - 3)) & 0x78u;
                            v[p] = *(lds_cf2v *)(uintptr_t)(ad + (uint32_t)(((g * 8 + p) * FP_EC + e) * 128));
                        }
#pragma unroll
                        for (int p = 0; p < 8; p += 2) {
                            if (g == 0 && p == 0) {          // (the sums start from the first terms, not from 0 + the first terms)
                                acc = v[0];
                                acc1 = v[1];
                            } else {
                                acc += v[p];
                                acc1 += v[p + 1];
                            }
                        }
                        __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);     // the addresses (VALU)
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // the reads (DS read)
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // the adds
                    } else {
                        for (int p = 0; p < NPAIR - 8 * g; ++p) {
                            const uint32_t off = ((zs >> (4 * p)) & 15u) << 3;
                            const fcd_f2v v = *(lds_cf2v *)(uintptr_t)(off + (uint32_t)(((g * 8 + p) * FP_EC + e) * 128));
                            acc += v;
                        }
                    }
                }
            }
            acc += acc1;
#ifdef FCD_ABLATE
            asm volatile("" ::"v"(acc.x), "v"(acc.y));
            const long long tc1 = clock64();
#endif
            const float4 ek = edge_k[e];
            if (FCD_ABL(0, 2)) {             // ablation: no RNG / exp
                f_state[((int64_t)w * C + c) * 64 + lane] = (uint8_t)(acc.x > acc.y ? 1 : 2);
                continue;
            }
            if ((e & 3) == 0) rnd = fcd_philox((uint32_t)(c >> 2), chain, sweep, FCD_KIND_F, k0, k1);   // c0 is a multiple of 8
            const uint32_t xw = fcd_word(rnd, e & 3);
            bool amb = false;
            int k;
            const float bf1 = ek.x + acc.x, bf2 = ek.y + acc.y, xf = (float)xw * 2.3283064e-10f;
            // (no exponential where the mode leads by more than e^15 in every lane -- the usual case on separated data -- unless
            // the test hook f_tol asks for the exact path everywhere)
            if (margin > 1.f || __ballot(!fcd_draw_f_sure(bf1, bf2, xf, ek.z, &k)) != 0ull) k = fcd_draw_f_fast32(bf1, bf2, xf, ek.z, margin, &amb);
            if (__ballot(amb) != 0ull) {
                // somewhere in the wave the fp32 sums cannot decide the draw: the edge again, in fp64, for the whole wave
                k = fcd_f_exact_edge(lMf + c * U * 6, lg1 + (S_B[c * 3 + 1] - S_B[c * 3 + 0]), lg2 + (S_B[c * 3 + 2] - S_B[c * 3 + 0]), Zc[0],
                                     NW16 > 1 ? Zc[NW16 > 1 ? 1 : 0] : 0u, NW16 > 2 ? Zc[NW16 > 2 ? 2 : 0] : 0u, NW16 > 3 ? Zc[NW16 > 3 ? 3 : 0] : 0u, U, xw);
                if (lane == 0) atomicAdd(dbg, 1ull);
            }
#ifdef FCD_ABLATE
            asm volatile("" ::"v"(k));
            const long long tc2 = clock64();
#endif
            (f_state + ((int64_t)w * C + c) * 64)[(uint32_t)lane] = (uint8_t)k;     // (scalar base + lane)
            if (fsq) {
                // square copy for the r pass that follows (fcd_gibbs_sweeps): rows of it are contiguous in m
                uint8_t *sq = fsq + (int64_t)w * Nreg * Nreg * 64;
                (sq + ((int64_t)wn * Nreg + wm) * 64)[(uint32_t)lane] = (uint8_t)k;
                (sq + ((int64_t)wm * Nreg + wn) * 64)[(uint32_t)lane] = (uint8_t)k;
            }
            if (e + 1 < FP_EC) {
#pragma unroll
                for (int j = 0; j < NW16; ++j) {
                    if (new_row) rn[j] = comp(vnn, j);
                    const uint32_t rm = comp(vmn, j);
                    Zc[j] = (rn[j] ^ rm) | ((rn[j] & rm) << 2);
                }
                wn = nn;
                wm = nm;
            }
#ifdef FCD_ABLATE
            {
                asm volatile("" ::"v"(Zc[0]));
                const long long tc3 = clock64();
                tr_terms += tc1 - tc0;
                tr_draw += tc2 - tc1;
                tr_rest += tc3 - tc2;
            }
#endif
        }
    }
#ifdef FCD_ABLATE
    FCD_TRACE_VAL(trec, 4, tr_terms);
    FCD_TRACE_VAL(trec, 5, tr_draw);
    FCD_TRACE_VAL(trec, 6, tr_rest);
#endif
    FCD_TRACE(trec, 3);
}

// ---------------------------------------------------------------------------------------------
// f step, pair form for ANY number of patients (what cfg5's U = 250 runs).  Same records, same term (one
// ds_read_b128 + two fp64 adds per PAIR of patients), but
//   * the tile holds EC edges with EC chosen by the host so that two workgroups share a CU (EC * NPAIR * 256 B <= 80 KiB),
//   * the pair records are built from 384-byte pieces of the rows that each wave stages in a small LDS scratch of its
//     own (no whole single rows in LDS: at U = 250 they would cost 12 KB per edge), every table byte loaded once,
//   * the slot words of a group of 16 patients are loaded per group, one group ahead (a wave cannot hold the
//     16 words per region that U = 250 needs), the group loop is a run-time loop.
// One Philox block still serves four edges; a tile of EC < 4 edges recomputes it (c0 is a multiple of EC, EC | 4).
// ---------------------------------------------------------------------------------------------
template <int EC>
__global__ __launch_bounds__(1024, 8) void gibbs_f_pairx_kernel(const double *__restrict__ S_B, const double *__restrict__ lMf,
                                                                const double *__restrict__ hyper, uint8_t *__restrict__ f_state,
                                                                const uint32_t *__restrict__ r_U, int Nreg, int U, int64_t C,
                                                                int GW, uint32_t chain0, uint64_t seed, uint32_t sweep,
                                                                float margin, uint8_t *__restrict__ fsq,
                                                                unsigned long long *__restrict__ dbg) {
    // pairs [EC][NPAIR][16] float2 | per-edge constants [8] float4 | staging scratch [waves][2][24] double2
    extern __shared__ __attribute__((aligned(256))) double ptile[];
    const int NPAIR = (U + 1) >> 1, NW16 = (U + 15) >> 4;
    const int64_t c0 = (int64_t)blockIdx.x * EC;
    const int ne = (int)((C - c0 < EC) ? (C - c0) : EC);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    int wn, wm;                          // (n, m) of the edge at hand (wave-uniform walk of the lower-triangular order)
    fcd_edge_to_pair(c0, wn, wm);
    const uint32_t *__restrict__ ru = r_U + (int64_t)(w < GW ? w : 0) * Nreg * (NW16 <= 4 ? 4 : NW16) * 64;
    const uint32_t ul = (uint32_t)lane;
    // slot words of the first group of the first edge: requested before the build so that their latency hides behind it
    // (slot word g of region n for this lane: [word][lane], or -- up to four words, forced onto this kernel by knob f_form = 2 --
    // the four side by side of ru_index)
    const uint32_t sn = NW16 <= 4 ? 256u : (uint32_t)NW16 * 64u, sg = NW16 <= 4 ? 1u : 64u, lo = NW16 <= 4 ? ul * 4u : ul;
    auto ruw = [&](int n, int g) -> uint32_t { return ru[(uint32_t)n * sn + (uint32_t)g * sg + lo]; };      // (no branch around a load)
    uint32_t rn_c = ruw(wn, 0), rm_c = ruw(wm, 0);
    float4 *edge_k = reinterpret_cast<float4 *>(reinterpret_cast<char *>(ptile) + (size_t)EC * NPAIR * 128);   // [8]
    if (threadIdx.x < EC) edge_k[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    {
        // Pair records from the tile's rows of lMf, every byte of the table loaded ONCE: a wave takes a group of four
        // pairs of one edge (8 patients = 384 contiguous bytes, fewer at the end of a row; missing patients read as
        // zero records), lanes 0..23 load one 16-byte value each into the wave's own LDS scratch, then every lane makes
        // the record entry of its (pair, slot) from two scratch values -- rounded ONCE to fp32.  Two groups per turn, their
        // loads in flight together.
        // (The 16 lanes of a pair loading their two operands straight from memory cost 2.9x the table's bytes in L2
        // misses at cfg5: the same line requested by several instructions in flight.)
        const int wv = (int)(threadIdx.x >> 6);
        double2 *scratch = reinterpret_cast<double2 *>(reinterpret_cast<char *>(edge_k) + 128) + wv * (2 * 24);
        fcd_f2v *dst = reinterpret_cast<fcd_f2v *>(ptile);
        const int slot = lane & 15, pl = lane >> 4;
        const int x0 = slot & 1, x1 = (slot >> 1) & 1, a0 = (slot >> 2) & 1, a1 = slot >> 3;
        const bool valid = !((x0 & a0) | (x1 & a1));
        const int l0 = a0 ? 1 : (x0 ? 2 : 0), l1 = a1 ? 1 : (x1 ? 2 : 0);          // 0 typical, 1 both, 2 discordant
        const int NG4 = (NPAIR + 3) >> 2;
        const int groups = ne * NG4, nwv = (int)(blockDim.x >> 6);
        const int row_d2 = U * 3;                                                   // 16-byte values per edge row
        const double2 *__restrict__ src = reinterpret_cast<const double2 *>(lMf + c0 * U * 6);
        for (int g0 = wv; g0 < groups; g0 += 2 * nwv) {
            double2 v[2];
            int eq[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int g = g0 + t * nwv;
                const int gc = g < groups ? g : groups - 1;
                const int e = gc / NG4, q = gc - e * NG4;
                eq[t][0] = e;
                eq[t][1] = q;
                const int i2 = q * 24 + lane;                                       // value index inside the row
                // (read once, by this workgroup only: non-temporal, so the stream does not push the r words out of L2)
                if (lane < 24 && i2 < row_d2) {
                    const double *p2 = reinterpret_cast<const double *>(src + (e * row_d2 + i2));
                    v[t] = make_double2(__builtin_nontemporal_load(p2), __builtin_nontemporal_load(p2 + 1));
                } else {
                    v[t] = make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (lane < 24) scratch[t * 24 + lane] = v[t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int g = g0 + t * nwv;
                const int pr = eq[t][1] * 4 + pl;
                const double2 va = scratch[t * 24 + pl * 6 + l0], vb = scratch[t * 24 + pl * 6 + 3 + l1];
                fcd_f2v r = {0.f, 0.f};
                if (valid) {
                    r.x = (float)(va.x + vb.x);
                    r.y = (float)(va.y + vb.y);
                }
                if (g < groups && pr < NPAIR) dst[(eq[t][0] * NPAIR + pr) * 16 + slot] = r;
                // the edge's bound B_e (sum over its patients of the largest |value| of the patient's row): lanes 0..7 take one
                // patient of the group each, the edge's sum gathers in LDS (any order: it only sets how cautious the draw is)
                if (g < groups && lane < 8) {
                    float a = 0.f;
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const double2 q = scratch[t * 24 + lane * 3 + l];
                        a = fmaxf(a, fmaxf((float)fabs(q.x), (float)fabs(q.y)));
                    }
                    atomicAdd(&edge_k[eq[t][0]].w, a * 1.0000002f);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                        // the scratch is free for the next turn
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();
    const double lg1 = hyper[FCD_H_LNGAMMA + 1] - hyper[FCD_H_LNGAMMA + 0];
    const double lg2 = hyper[FCD_H_LNGAMMA + 2] - hyper[FCD_H_LNGAMMA + 0];
    if ((int)threadIdx.x < ne) {
        // the edge's constants as the draw wants them: the two log-odds offsets in fp32 and the error bound of the whole
        // fp32 sum (NPAIR records + the offset + the initial zero)
        const int64_t c = c0 + threadIdx.x;
        const double c1 = lg1 + (S_B[c * 3 + 1] - S_B[c * 3 + 0]), c2 = lg2 + (S_B[c * 3 + 2] - S_B[c * 3 + 0]);
        const float cm = fmaxf((float)fabs(c1), (float)fabs(c2)) * 1.0000002f;
        const float a = edge_k[threadIdx.x].w;
        edge_k[threadIdx.x] = make_float4((float)c1, (float)c2, fcd_draw_f_eta(fcd_f32_sum_err(NPAIR + 1, a) + fcd_f32_offset_err(cm, a)), 0.f);
    }
    __syncthreads();
    if (w >= GW) return;
    if (FCD_ABL(0, 3)) return;           // ablation: staging and build only
    const uint32_t chain = chain0 + (uint32_t)w * 64u + lane;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    fcd_u4 rnd = {0, 0, 0, 0};
    const int NG = (NPAIR + 7) >> 3;     // groups of 8 pairs = 16 patients = one slot word
    const uint32_t tile_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ptile;   // 256-aligned

    for (int e = 0; e < ne; ++e) {
        const int64_t c = c0 + e;
        // next edge: (n, m+1), or (n+1, 0) at the end of row n (clamped past the last edge: unused there)
        int nn = wn, nm = wm + 1;
        if (nm == nn) {
            nm = 0;
            nn = (nn + 1 < Nreg) ? nn + 1 : nn;
        }
        const uint32_t tb = __builtin_amdgcn_readfirstlane(tile_off + (uint32_t)(e * NPAIR * 128));
        fcd_f2v acc = {0.f, 0.f}, acc1 = {0.f, 0.f};    // even / odd pairs: two chains of packed adds
        for (int g = 0; g < NG; ++g) {
            // slot words of the next group (of the next edge after the last group): always loaded, from a valid place
            const bool last = g + 1 == NG;
            const int pn = last ? nn : wn, pm = last ? nm : wm, pg = last ? 0 : g + 1;
            const uint32_t rn_n = ruw(pn, pg), rm_n = ruw(pm, pg);
            const uint32_t zs = (rn_c ^ rm_c) | ((rn_c & rm_c) << 2);
            const uint32_t gb = tb + (uint32_t)(g * (8 * 128));
            if (NPAIR - 8 * g >= 8) {
                fcd_f2v v[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const uint32_t sh = (p == 0) ? (zs << 3) : (zs >> (4 * p - 3));
                    uint32_t ad;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(sh), "s"(0x78u), "v"(gb));
                    v[p] = *(lds_cf2v *)(uintptr_t)(ad + (uint32_t)(p * 128));
                }
#pragma unroll
                for (int p = 0; p < 8; p += 2) {
                    acc += v[p];
                    acc1 += v[p + 1];
                }
            } else {
                for (int p = 0; p < NPAIR - 8 * g; ++p) {
                    const uint32_t off = ((zs >> (4 * p)) & 15u) << 3;
                    acc += *(lds_cf2v *)(uintptr_t)(off + gb + (uint32_t)(p * 128));
                }
            }
            rn_c = rn_n;
            rm_c = rm_n;
        }
        acc += acc1;
        const float4 ek = edge_k[e];
        if (e == 0 || (c & 3) == 0) rnd = fcd_philox((uint32_t)(c >> 2), chain, sweep, FCD_KIND_F, k0, k1);
        const uint32_t xw = fcd_word(rnd, (int)(c & 3));
        bool amb = false;
        int k;
        const float bf1 = ek.x + acc.x, bf2 = ek.y + acc.y, xf = (float)xw * 2.3283064e-10f;
        if (margin > 1.f || __ballot(!fcd_draw_f_sure(bf1, bf2, xf, ek.z, &k)) != 0ull) k = fcd_draw_f_fast32(bf1, bf2, xf, ek.z, margin, &amb);
        if (__ballot(amb) != 0ull) {
            // somewhere in the wave the fp32 sums cannot decide the draw: the edge again, in fp64, for the whole wave
            bool first = true;
            double b1 = 0.0, b2 = 0.0;
            const double *row = lMf + c * U * 6;
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                const uint32_t a_ = ruw(wn, g), b_ = ruw(wm, g);
                fcd_f_exact_group(row, U, 16 * g, min(8, NPAIR - 8 * g), (a_ ^ b_) | ((a_ & b_) << 2), first, b1, b2);
            }
            b1 = lg1 + ((S_B[c * 3 + 1] - S_B[c * 3 + 0]) + b1);
            b2 = lg2 + ((S_B[c * 3 + 2] - S_B[c * 3 + 0]) + b2);
            k = fcd_draw_f(0.0, b1, b2, fcd_u32(xw));
            if (lane == 0) atomicAdd(dbg, 1ull);
        }
        (f_state + ((int64_t)w * C + c) * 64)[(uint32_t)lane] = (uint8_t)k;     // (scalar base + lane)
        if (fsq) {
            uint8_t *sq = fsq + (int64_t)w * Nreg * Nreg * 64;
            (sq + ((int64_t)wn * Nreg + wm) * 64)[(uint32_t)lane] = (uint8_t)k;
            (sq + ((int64_t)wm * Nreg + wn) * 64)[(uint32_t)lane] = (uint8_t)k;
        }
        wn = nn;
        wm = nm;
    }
}

// conditional log-weights of every r site given the CURRENT state (nothing updated): parity hook.
__global__ __launch_bounds__(64) void gibbs_cond_r_kernel(const double *__restrict__ lM, const double *__restrict__ hyper,
                                                          const uint8_t *__restrict__ f_state,
                                                          const uint64_t *__restrict__ r_bits, int Nreg, int U, int64_t C,
                                                          int64_t G, int mode, double *__restrict__ cond_r) {
    const int u = blockIdx.x, w = blockIdx.y, n = blockIdx.z, lane = threadIdx.x;
    const uint8_t *fw = f_state + (int64_t)w * C * 64 + lane;
    double s0 = 0.0, s1 = 0.0;
    for (int m = 0; m < Nreg; ++m) {
        if (m == n) continue;
        const int64_t c = fcd_pair_to_edge(n, m, mode);
        const int k = fw[c * 64];
        const uint32_t bit = (uint32_t)((r_bits[((int64_t)w * Nreg + m) * U + u] >> lane) & 1ull);
        const double *p = lM + (c * U + u) * 9 + k * 3;
        s0 += p[bit * 2];
        s1 += p[2 - bit];
    }
    const int64_t g = (int64_t)w * 64 + lane;
    if (g < G) {
        double *o = cond_r + ((g * Nreg + n) * U + u) * 2;
        o[0] = hyper[FCD_H_LNPI0] + s0;
        o[1] = hyper[FCD_H_LNPI1] + s1;
    }
}

// ---------------------------------------------------------------------------------------------
// pooled statistics, marginal counters, M-step
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gibbs_stats_kernel(const uint8_t *__restrict__ f_state,
                                                          const uint64_t *__restrict__ r_bits, int64_t C, int64_t NU, int GW,
                                                          int64_t G, unsigned long long *__restrict__ counts) {
    __shared__ unsigned long long red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long cr = 0, c0 = 0, c1 = 0, c2 = 0;
    // f: one wave per (w, c) row of 64 bytes
    const int64_t rows = (int64_t)GW * C;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int w = (int)(row / C);
        const uint64_t act = fcd_active_mask(w, G);
        const int f = f_state[row * 64 + lane];
        const uint64_t b0 = __ballot(f == 0) & act, b1 = __ballot(f == 1) & act, b2 = __ballot(f == 2) & act;
        if (lane == 0) {
            c0 += __popcll(b0);
            c1 += __popcll(b1);
            c2 += __popcll(b2);
        }
    }
    // r: one thread per word
    const int64_t words = (int64_t)GW * NU;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i / NU);
        cr += __popcll(r_bits[i] & fcd_active_mask(w, G));
    }
    // integer sums: any order gives the same result
    for (int o = 32; o > 0; o >>= 1) cr += __shfl_xor(cr, o, 64);
    if (lane == 0) {
        red[wave][0] = cr; red[wave][1] = c0; red[wave][2] = c1; red[wave][3] = c2;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (s) atomicAdd(&counts[threadIdx.x], s);
    }
    if (blockIdx.x == 0 && threadIdx.x == 4) counts[4] = (unsigned long long)G;
}

// (pi, gamma) from pooled counts {sum r, #f=0, #f=1, #f=2, chains}: the sample version of fit.py:208-220
__device__ __forceinline__ void mstep_from_counts(const unsigned long long *c5, double sites_per_chain_r, double C,
                                                  double *__restrict__ hyper) {
    const double Gtot = (double)c5[4];
    const double n_r = Gtot * sites_per_chain_r;
    double pi = (double)c5[0] / n_r;                      // fit.py:213 over chains
    const double lo = 0.5 / n_r;
    pi = fmin(fmax(pi, lo), 1.0 - lo);
    hyper[FCD_H_LNPI0] = log(1.0 - pi);
    hyper[FCD_H_LNPI1] = log(pi);
    const double n_f = Gtot * C;
    for (int k = 0; k < 3; ++k) {
        double g = (double)c5[1 + k] / n_f;               // fit.py:220 over chains
        g = fmax(g, 0.5 / n_f);
        hyper[FCD_H_LNGAMMA + k] = log(g);
    }
}

__global__ void gibbs_mstep_kernel(const long long *__restrict__ counts, double sites_per_chain_r, double C,
                                   double *__restrict__ hyper) {
    unsigned long long c5[5];
    for (int i = 0; i < 5; ++i) c5[i] = (unsigned long long)counts[i];
    mstep_from_counts(c5, sites_per_chain_r, C, hyper);
}

__global__ __launch_bounds__(256) void gibbs_accum_kernel(const uint8_t *__restrict__ f_state,
                                                          const uint64_t *__restrict__ r_bits, int64_t C, int64_t NU, int GW,
                                                          int64_t G, uint32_t *__restrict__ cnt_f, uint32_t *__restrict__ cnt_r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < C; c += (int64_t)gridDim.x * 4) {
        uint32_t n0 = 0, n1 = 0, n2 = 0;
        for (int w = 0; w < GW; ++w) {
            const uint64_t act = fcd_active_mask(w, G);
            const int f = f_state[((int64_t)w * C + c) * 64 + lane];
            n0 += __popcll(__ballot(f == 0) & act);
            n1 += __popcll(__ballot(f == 1) & act);
            n2 += __popcll(__ballot(f == 2) & act);
        }
        if (lane == 0) {
            cnt_f[c * 3 + 0] += n0;
            cnt_f[c * 3 + 1] += n1;
            cnt_f[c * 3 + 2] += n2;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < NU; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t s = 0;
        for (int w = 0; w < GW; ++w) s += __popcll(r_bits[(int64_t)w * NU + i] & fcd_active_mask(w, G));
        cnt_r[i] += s;
    }
}

// ---------------------------------------------------------------------------------------------
// tally: pooled counts AND marginal counters in one pass over the state (f_state is read once, 16 bytes
// per lane: a wave covers the 16 chain words x 64 chains of an edge with a single load instruction) -- and the
// odds and ends that used to be launches of their own around it:
//   * the pooled counts are summed in accumulators that belong to the context (acc[0..3] + a ticket in acc[4], zero
//     between launches): the block that draws the last ticket reads the totals, writes counts_out[0..4], runs the
//     (pi, gamma) M-step when hyper != nullptr (one rank: the statistics need no exchange) and puts the accumulators
//     back to zero -- no memset launch, no one-thread M-step launch;
//   * r_U != nullptr: the per-lane slot words of the NEXT f pass (pack_ru_kernel's job) are made here from the r
//     bits this kernel reads anyway.
// Integer sums: any order gives the same totals.
// ---------------------------------------------------------------------------------------------
struct tally_args {
    const uint8_t *f_state;
    const uint64_t *r_bits;
    int64_t C, NU, G;
    int GW, Nreg, U, NW;               // (NW slot-source words per region: pack_ru_word)
    unsigned long long *acc;           // context-owned: 4 sums + ticket
    unsigned long long *counts_out;    // nullable
    uint32_t *cnt_f, *cnt_r;           // nullable (both or neither)
    double *hyper;                     // nullable: M-step target
    uint32_t *r_U;                     // nullable: slot words of the next f pass
    int n_f_blocks;                    // blocks that read the f state (0: done elsewhere)
    int n_r_blocks;                    // blocks that count the r bits (at least; the f blocks do too); the rest make r_U
};

__global__ __launch_bounds__(1024) void gibbs_tally_kernel(const tally_args a) {
    __shared__ unsigned long long red[16], redf[16][3];
    __shared__ int sh_last;
    const uint8_t *__restrict__ f_state = a.f_state;
    const uint64_t *__restrict__ r_bits = a.r_bits;
    uint32_t *__restrict__ cnt_f = a.cnt_f;
    uint32_t *__restrict__ cnt_r = a.cnt_r;
    const int64_t C = a.C, NU = a.NU, G = a.G;
    const int GW = a.GW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blocks [0, nfb): the f state (nfb == 0: the r pass's packing launch has seen to it) -- and the r bits; blocks
    // [nfb, gridDim): the slot words of the next f pass (they run beside the others on CUs of their own instead of after them)
    const int nfb = a.n_f_blocks;
    const bool f_role = (int)blockIdx.x < nfb;
    if (f_role) {
        fcd_tally_f tf;
        tf.f_state = f_state; tf.C = C; tf.G = G; tf.GW = GW; tf.acc = a.acc; tf.cnt_f = cnt_f;
        fcd_tally_f_block<16>(tf, (int)blockIdx.x, nfb, redf);
    }
    // the r bits: every block of the first max(nfb, n_r_blocks) takes its share
    const int nrb = nfb > a.n_r_blocks ? nfb : a.n_r_blocks;
    const bool r_role = (int)blockIdx.x < nrb;
    unsigned long long cr = 0;
    for (int64_t i = r_role ? (int64_t)blockIdx.x * blockDim.x + threadIdx.x : NU; i < NU; i += (int64_t)nrb * blockDim.x) {
        uint32_t sr = 0;
        for (int w = 0; w < GW; ++w) sr += __popcll(r_bits[(int64_t)w * NU + i] & fcd_active_mask(w, G));
        if (cnt_r) atomicAdd(&cnt_r[i], sr);
        cr += sr;
    }
    if (a.r_U && (int)blockIdx.x >= nrb) {
        // slot words of the next f pass: one wave per (w, n, word) item, as pack_ru_kernel
        const int U = a.U, NW = a.NW;
        const int items = GW * a.Nreg * NW;
        for (int item = ((int)blockIdx.x - nrb) * 16 + wave; item < items; item += ((int)gridDim.x - nrb) * 16) {
            const int jw = item % NW, wn = item / NW;                   // wn = w*Nreg + n
            a.r_U[ru_index(wn, NW, jw, lane)] = pack_ru_word(r_bits, wn, U, jw, lane);
        }
    }
    if (!a.acc) return;
    // Only the blocks that count take a ticket: the sums, the ticket and the M-step -- a chain of four trips to the memory
    // side -- then run BESIDE the blocks that make the slot words instead of behind the last of them.
    if (!r_role) return;
    // the r count: one atomic per BLOCK (same-address atomics serialise): wave sums -> LDS -> thread 0
    for (int o = 32; o > 0; o >>= 1) cr += __shfl_xor(cr, o, 64);
    if (lane == 0) red[wave] = cr;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int q = 0; q < 16; ++q) t += red[q];
        if (t) {
            // with the old value asked for, the add has been performed at the memory side once it returns: the
            // barrier below then orders it before this block's ticket (every access to acc[] is a device-scope atomic)
            const unsigned long long old = atomicAdd(&a.acc[0], t);
            asm volatile("" ::"v"(old));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sh_last = (atomicAdd(&a.acc[4], 1ull) == (unsigned long long)nrb - 1ull) ? 1 : 0;
    __syncthreads();
    if (sh_last && threadIdx.x < 64) {
        // the five words are fetched (and reset) by five lanes at once: one memory round trip, not five in a row
        unsigned long long mine = 0;
        if (threadIdx.x < 5) mine = atomicExch(&a.acc[threadIdx.x], 0ull);     // read at the memory side and reset
        unsigned long long c5[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) c5[i] = __shfl(mine, i, 64);
        c5[4] = (unsigned long long)G;
        if (a.counts_out && threadIdx.x < 8) a.counts_out[threadIdx.x] = threadIdx.x < 5 ? c5[threadIdx.x] : 0ull;
        if (a.hyper && threadIdx.x < 5) {
            // the M-step of mstep_from_counts, one logarithm per lane: lanes 0..2 ln gamma_k, lane 3 ln(1 - pi), lane 4 ln pi
            const double Gtot = (double)c5[4];
            const int i = (int)threadIdx.x;
            double v;
            if (i < 3) {
                const double n_f = Gtot * (double)C;
                v = fmax((double)c5[1 + i] / n_f, 0.5 / n_f);             // fit.py:220 over chains
            } else {
                const double n_r = Gtot * (double)NU;
                const double lo = 0.5 / n_r;
                const double pi = fmin(fmax((double)c5[0] / n_r, lo), 1.0 - lo);      // fit.py:213 over chains
                v = i == 3 ? 1.0 - pi : pi;
            }
            a.hyper[i < 3 ? FCD_H_LNGAMMA + i : (i == 3 ? FCD_H_LNPI0 : FCD_H_LNPI1)] = log(v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// log-joint per chain: grid = (GW, LJ_SLICES); slice s sums edges c = s, s+LJ_SLICES, ... and (slice 0) the
// prior of r; partial[s][g] in the workspace, folded in slice order.
// ---------------------------------------------------------------------------------------------
constexpr int LJ_SLICES = 64;
__global__ __launch_bounds__(64) void gibbs_logjoint_kernel(const double *__restrict__ S_B, const double *__restrict__ lM,
                                                            const double *__restrict__ hyper,
                                                            const uint8_t *__restrict__ f_state,
                                                            const uint64_t *__restrict__ r_bits, int Nreg, int U, int64_t C,
                                                            int GW, double *__restrict__ partial) {
    const int w = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    const uint64_t *rw = r_bits + (int64_t)w * Nreg * U;
    double acc = 0.0;
    for (int64_t c = s; c < C; c += LJ_SLICES) {
        int n, m;
        fcd_edge_to_pair(c, n, m);
        const int k = f_state[((int64_t)w * C + c) * 64 + lane];
        double e = hyper[FCD_H_LNGAMMA + k] + S_B[c * 3 + k];
        for (int u = 0; u < U; ++u) {
            const uint32_t a = (uint32_t)((rw[(int64_t)n * U + u] >> lane) & 1ull);
            const uint32_t b2 = (uint32_t)((rw[(int64_t)m * U + u] >> lane) & 1ull);
            const int l = (a & b2) ? 1 : ((a ^ b2) ? 2 : 0);
            e += lM[(c * U + u) * 9 + k * 3 + l];
        }
        acc += e;
    }
    if (s == 0) {
        int ones = 0;
        for (int64_t i = 0; i < (int64_t)Nreg * U; ++i) ones += (int)((rw[i] >> lane) & 1ull);
        acc += (double)ones * hyper[FCD_H_LNPI1] + (double)((int64_t)Nreg * U - ones) * hyper[FCD_H_LNPI0];
    }
    partial[((int64_t)s * GW + w) * 64 + lane] = acc;
}

// per-chain number of anomalous (region, patient) sites: sum_{n,u} r_nu -- the scalar SURVEY.md section 8f item 4 names
// for the chain diagnostics next to the log-joint.  One wave per (chain word, slice of the sites); out is zeroed first.
__global__ __launch_bounds__(64) void gibbs_rsum_kernel(const uint64_t *__restrict__ r_bits, int64_t NU, int64_t G,
                                                        unsigned int *__restrict__ out) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const uint64_t *rw = r_bits + (int64_t)w * NU;
    unsigned int ones = 0;
    for (int64_t i = blockIdx.y; i < NU; i += gridDim.y) ones += (unsigned int)((rw[i] >> lane) & 1ull);
    const int64_t g = (int64_t)w * 64 + lane;
    if (g < G && ones) atomicAdd(&out[g], ones);
}

__global__ void gibbs_logjoint_fold(const double *__restrict__ partial, int64_t GWx64, int64_t G, double *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double acc = 0.0;
    for (int s = 0; s < LJ_SLICES; ++s) acc += partial[(int64_t)s * GWx64 + g];
    out[g] = acc;
}

// ---------------------------------------------------------------------------------------------
// packed <-> plain state
// ---------------------------------------------------------------------------------------------
__global__ void export_f_kernel(const uint8_t *__restrict__ f_state, int64_t C, int64_t G, uint8_t *__restrict__ f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over G*C, c fastest
    if (i >= G * C) return;
    const int64_t g = i / C, c = i % C;
    f[i] = f_state[((g >> 6) * C + c) * 64 + (g & 63)];
}
__global__ void export_r_kernel(const uint64_t *__restrict__ r_bits, int64_t NU, int64_t G, uint8_t *__restrict__ r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over G*NU
    if (i >= G * NU) return;
    const int64_t g = i / NU, j = i % NU;
    r[i] = (uint8_t)((r_bits[(g >> 6) * NU + j] >> (g & 63)) & 1ull);
}
__global__ void import_f_kernel(const uint8_t *__restrict__ f, int64_t C, int64_t G, int GW, uint8_t *__restrict__ f_state) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over GW*C*64
    if (i >= (int64_t)GW * C * 64) return;
    const int64_t lane = i & 63, c = (i >> 6) % C, w = (i >> 6) / C;
    const int64_t g = w * 64 + lane;
    f_state[i] = g < G ? f[g * C + c] : 0;
}
__global__ __launch_bounds__(256) void import_r_kernel(const uint8_t *__restrict__ r, int64_t NU, int64_t G, int GW,
                                                       uint64_t *__restrict__ r_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t word = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // over GW*NU
    if (word >= (int64_t)GW * NU) return;
    const int64_t w = word / NU, j = word % NU;
    const int64_t g = w * 64 + lane;
    const uint64_t ball = __ballot(g < G ? (r[g * NU + j] != 0) : false);
    if (lane == 0) r_bits[word] = ball;
}

__global__ void philox_uniforms_kernel(const uint32_t *__restrict__ ctr4, int64_t n, uint64_t seed, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fcd_u4 x = fcd_philox(ctr4[i * 4 + 0], ctr4[i * 4 + 1], ctr4[i * 4 + 2], ctr4[i * 4 + 3], (uint32_t)seed,
                                (uint32_t)(seed >> 32));
    out[i * 2 + 0] = fcd_u53(x.x, x.y);
    out[i * 2 + 1] = fcd_u53(x.z, x.w);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// edges per LDS tile of the f step and waves per block
void f_step_geometry(int64_t U, int GW, int &Ec, int &wpb, size_t &shmem) {
    wpb = GW < 16 ? GW : 16;
    const size_t per_edge = (size_t)U * 72;
    int64_t e = (int64_t)(32 * 1024 / per_edge);
    if (e < 1) e = 1;
    if (e > 8) e = 8;
    if (e > 1) e &= ~1ll;  // even: both halves of a Philox block are used inside one tile
    Ec = (int)e;
    shmem = per_edge * Ec;
}

template <bool COND>
int launch_f(fcd_ctx *ctx, const double *S_B, const double *lM, const double *hyper, uint8_t *f_state,
             const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, const fcd_geo &g, int64_t chain0, uint64_t seed,
             int64_t sweep, double *cond_f, hipStream_t s) {
    int Ec, wpb;
    size_t shmem;
    f_step_geometry(U, g.GW, Ec, wpb, shmem);
    if (shmem > 160 * 1024)
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "f step: one edge's table row (U=%lld patients) exceeds the 160 KiB LDS", U);
    {
        int rc = fcd_lds_attr(ctx, COND ? FCD_KA_F_COND : FCD_KA_F_GENERIC, reinterpret_cast<const void *>(&gibbs_f_kernel<COND>), shmem);
        if (rc) return rc;
    }
    dim3 grid((unsigned)((g.C + Ec - 1) / Ec), (unsigned)((g.GW + wpb - 1) / wpb));
    hipLaunchKernelGGL(gibbs_f_kernel<COND>, grid, dim3(64 * wpb), shmem, s, S_B, lM, hyper, f_state, r_bits, (int)Nreg,
                       (int)U, g.C, g.GW, G, Ec, (uint32_t)chain0, seed, (uint32_t)sweep, cond_f);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

}  // namespace

extern "C" int fcd_gibbs_state_size(int64_t Nreg, int64_t U, int64_t G, size_t *f_bytes, size_t *r_bytes) {
    if (Nreg < 2 || U < 1 || G < 1 || !f_bytes || !r_bytes) return FCD_ERR_ARG;
    const int64_t GW = (G + 63) / 64;
    *f_bytes = (size_t)(GW * fcd_tri(Nreg) * 64);
    *r_bytes = (size_t)(GW * Nreg * U * 8);
    return FCD_OK;
}

extern "C" int fcd_gibbs_init(fcd_ctx *ctx, uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                              int64_t chain0, uint64_t seed, double pi, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, chain0, g);
    if (rc) return rc;
    if (!f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_init: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t items_f = (g.C + 1) / 2 * g.GW;
    hipLaunchKernelGGL(gibbs_init_f, dim3((unsigned)((items_f + 3) / 4)), dim3(256), 0, s, f_state, g.C, g.GW,
                       (uint32_t)chain0, seed);
    FCD_LAUNCH_CHECK();
    const int64_t items_r = (Nreg + 1) / 2 * U * g.GW;
    hipLaunchKernelGGL(gibbs_init_r, dim3((unsigned)((items_r + 3) / 4)), dim3(256), 0, s, r_bits, (int)Nreg, (int)U, g.GW,
                       (uint32_t)chain0, seed, pi);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_edge_tables(fcd_ctx *ctx, const double *lM, int64_t Nreg, int64_t U, double *lMf,
                                     fcd_stream stream) {
    if (!ctx || !lM || !lMf) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_edge_tables: null pointer");
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    const int64_t n_items = fcd_tri(Nreg) * U;
    int64_t blocks = (n_items + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(edge_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lM, n_items, lMf);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_f_step(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *hyper,
                                uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                                int64_t chain0, uint64_t seed, int64_t sweep, fcd_stream stream) {
    return fcd_gibbs_f_step_sq(ctx, S_B, lM, lMf, hyper, f_state, r_bits, Nreg, U, G, chain0, seed, sweep, (hipStream_t)stream,
                               nullptr, false);
}

// Which kernel the f step runs at a shape, and with what geometry: shared by the launch code, by the workspace
// formula (fcd_f_pass_ws_bytes) and by the fused driver (only the pair forms write the square copy).
enum { F_GENERIC = 0, F_PAIR = 1, F_PAIRX = 2, F_DIFF = 3 };
struct f_plan {
    int form, NW16, EC;
    size_t shmem;
    int NW;                 // slot-source words per region (pack_ru_word): what pack_ru / the tally make
};
static f_plan f_plan_for(bool have_lMf, int64_t Nreg, int64_t U, int64_t GW, int f_form) {
    f_plan p = {F_GENERIC, (int)((U + 15) / 16), 0, 0, (int)((U + 15) / 16)};
    if (!have_lMf || (size_t)U * 48 > 160 * 1024) return p;
    const bool words_ok = GW * Nreg * p.NW16 < INT32_MAX / 4;       // r_U item index in 32 bits
    const size_t pair_shmem = (size_t)FP_EC * ((U + 1) / 2) * 128 + FP_EC * 16 + (size_t)FP_EC * U * 48;   // fp32 records, edge constants, fp64 rows
    if (f_form != F_PAIRX && f_form != F_DIFF && p.NW16 <= 4 && pair_shmem <= 96 * 1024 && words_ok) {
        p.form = F_PAIR;
        p.EC = FP_EC;
        p.shmem = pair_shmem;
        return p;
    }
    const size_t per_edge = (size_t)((U + 1) / 2) * 128, extra = 128 + 16 * 48 * 16;   // fp32 records; edge constants + a 768-byte staging scratch per wave
    if (f_form != F_DIFF && words_ok && per_edge + extra <= 160 * 1024) {
        // largest tile that still lets two workgroups share a CU; one edge per tile may take the whole LDS
        int ec = 8;
        while (ec > 1 && (size_t)ec * per_edge + extra > 80 * 1024) ec >>= 1;
        p.form = F_PAIRX;
        p.EC = ec;
        p.shmem = (size_t)ec * per_edge + extra;
        return p;
    }
    p.form = F_DIFF;
    int64_t e = (int64_t)(24 * 1024 / ((size_t)U * 48));
    if (e < 1) e = 1;
    if (e > 8) e = 8;
    if (e > 1) e &= ~1ll;   // even: both halves of a Philox block are used inside one tile
    p.EC = (int)e;
    p.shmem = (size_t)U * 48 * e;
    return p;
}

size_t fcd_f_pass_ws_bytes(int64_t Nreg, int64_t U, int64_t GW) {
    // per-lane slot words r_U of the pair forms (the largest user; the other forms need nothing): 16 patients per word
    // (up to four words side by side per lane, else [word][lane]: ru_index)
    const int64_t NW = (U + 15) / 16;
    return (size_t)GW * Nreg * (NW <= 4 ? 4 : NW) * 64 * sizeof(uint32_t);
}

size_t fcd_fsq_need_bytes(int64_t Nreg, int64_t U, int64_t GW) {
    const f_plan p = f_plan_for(true, Nreg, U, GW, 0);
    const size_t need = (size_t)GW * Nreg * Nreg * 64;
    return ((p.form == F_PAIR || p.form == F_PAIRX) && need <= ((size_t)8 << 30)) ? need : 0;
}

int fcd_gibbs_f_step_sq(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *hyper,
                        uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0,
                        uint64_t seed, int64_t sweep, hipStream_t stream, uint8_t *fsq, bool ru_ready, size_t ru_off) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, chain0, g);
    if (rc) return rc;
    if (!S_B || !lM || !hyper || !f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_f_step: null pointer");
    const f_plan pl = f_plan_for(lMf != nullptr, Nreg, U, g.GW, ctx->knobs.f_form);
    if (pl.form == F_GENERIC)
        return launch_f<false>(ctx, S_B, lM, hyper, f_state, r_bits, Nreg, U, G, g, chain0, seed, sweep, nullptr,
                               (hipStream_t)stream);
    fcd_abl_refresh((hipStream_t)stream);
    const int wpb = g.GW < 16 ? g.GW : 16;
    hipStream_t s = (hipStream_t)stream;
    float margin = FCD_DRAW_F_MARGIN;
    if ((float)ctx->knobs.f_tol > margin) margin = (float)ctx->knobs.f_tol;   // test hook: huge = every draw through fcd_draw_f
    if (pl.form == F_PAIR || pl.form == F_PAIRX) {
        // pair / triple forms: per-lane slot words over patients (scratch in the ctx workspace), records in LDS
        rc = fcd_ws_reserve(ctx, ru_off + fcd_f_pass_ws_bytes(Nreg, U, g.GW));
        if (rc) return rc;
        uint32_t *r_U = (uint32_t *)((char *)ctx->ws + ru_off);
        if (!ru_ready) {     // (inside fcd_gibbs_run the previous sweep's tally has made them already)
            const int64_t items = (int64_t)g.GW * Nreg * pl.NW;
            hipLaunchKernelGGL(pack_ru_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, r_bits, (int)Nreg, (int)U, pl.NW,
                               g.GW, r_U);
            FCD_LAUNCH_CHECK();
        }
        dim3 grid((unsigned)((g.C + pl.EC - 1) / pl.EC), (unsigned)((g.GW + wpb - 1) / wpb));
#define FCD_F_ARGS S_B, lMf, hyper, f_state, r_U, (int)Nreg, (int)U, g.C, g.GW, (uint32_t)chain0, seed, (uint32_t)sweep, margin, fsq, (unsigned long long *)ctx->dbg
#define FCD_LAUNCH_F(KERN, SLOT)                                                                              \
    do {                                                                                                      \
        rc = fcd_lds_attr(ctx, SLOT, reinterpret_cast<const void *>(&KERN), pl.shmem);                        \
        if (rc) return rc;                                                                                    \
        {                                                                                                     \
            static int lds0 = 0; /* (the pair form's record addresses start at LDS address 0) */              \
            rc = fcd_static_lds_check(ctx, reinterpret_cast<const void *>(&KERN), &lds0);                     \
            if (rc) return rc;                                                                                \
        }                                                                                                     \
        fcd_prof_begin(ctx, FCD_PROF_F, s);                                                                   \
        hipLaunchKernelGGL(KERN, grid, dim3(64 * wpb), pl.shmem, s, FCD_F_ARGS);                              \
        fcd_prof_end(ctx, FCD_PROF_F, s);                                                                     \
    } while (0)
        if (pl.form == F_PAIR) {
            if (pl.NW16 == 1) FCD_LAUNCH_F(gibbs_f_pair_kernel<1>, FCD_KA_F_PAIR + 0);
            else if (pl.NW16 == 2) FCD_LAUNCH_F(gibbs_f_pair_kernel<2>, FCD_KA_F_PAIR + 1);
            else if (pl.NW16 == 3) FCD_LAUNCH_F(gibbs_f_pair_kernel<3>, FCD_KA_F_PAIR + 2);
            else FCD_LAUNCH_F(gibbs_f_pair_kernel<4>, FCD_KA_F_PAIR + 3);
        } else {
            if (pl.EC == 8) FCD_LAUNCH_F(gibbs_f_pairx_kernel<8>, FCD_KA_F_PAIR_BIG + 0);
            else if (pl.EC == 4) FCD_LAUNCH_F(gibbs_f_pairx_kernel<4>, FCD_KA_F_PAIR_BIG + 1);
            else if (pl.EC == 2) FCD_LAUNCH_F(gibbs_f_pairx_kernel<2>, FCD_KA_F_PAIR_BIG + 2);
            else FCD_LAUNCH_F(gibbs_f_pairx_kernel<1>, FCD_KA_F_PAIR_BIG + 3);
        }
#undef FCD_LAUNCH_F
#undef FCD_F_ARGS
        FCD_LAUNCH_CHECK();
        return FCD_OK;
    }
    // a patient row that fits no pair tile (U > 1280): log-odds form with scalar r masks
    rc = fcd_lds_attr(ctx, FCD_KA_F_DIFF, reinterpret_cast<const void *>(&gibbs_f_diff_kernel), pl.shmem);
    if (rc) return rc;
    dim3 grid((unsigned)((g.C + pl.EC - 1) / pl.EC), (unsigned)((g.GW + wpb - 1) / wpb));
    fcd_prof_begin(ctx, FCD_PROF_F, s);
    hipLaunchKernelGGL(gibbs_f_diff_kernel, grid, dim3(64 * wpb), pl.shmem, s, S_B, lMf, hyper, f_state, r_bits,
                       (int)Nreg, (int)U, g.C, g.GW, pl.EC, (uint32_t)chain0, seed, (uint32_t)sweep, margin);
    fcd_prof_end(ctx, FCD_PROF_F, s);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_stats(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                               int64_t G, int64_t *counts, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!f_state || !r_bits || !counts) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_stats: null pointer");
    hipStream_t s = (hipStream_t)stream;
    FCD_HIP_TRY(hipMemsetAsync(counts, 0, 8 * sizeof(int64_t), s));
    int64_t blocks = ((int64_t)g.GW * g.C + 3) / 4;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(gibbs_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, f_state, r_bits, g.C, Nreg * U, g.GW, G,
                       reinterpret_cast<unsigned long long *>(counts));
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_sweeps(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *lMd,
                                const double *hyper,
                                uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0,
                                uint64_t seed, int64_t sweep0, int64_t n_sweeps, int edge_mode, int64_t *counts,
                                fcd_stream stream) {
    if (n_sweeps < 0 || sweep0 < 0 || sweep0 + n_sweeps > (1ll << 32))
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_sweeps: sweep range [%lld, +%lld) outside the 32-bit counter word", sweep0, n_sweeps);
    // symmetric edge ids + a pair-form f kernel + the blocked r pass: the f pass leaves a square copy of the f state
    // from which the r pass packs its f words (contiguous rows instead of 64-byte gathers: ~40 us -> ~10 us at cfg3).
    // The copy lives in the context (fcd_ctx_reserve sizes it; a first call at a larger shape grows it here, which
    // synchronises -- the one case the header names).
    uint8_t *fsq = nullptr;
    if (ctx && lMf && lMd && edge_mode == FCD_EDGE_SYMMETRIC && Nreg >= 2 && U >= 1 && G >= 1 && true) {
        const int64_t GW = (G + 63) / 64;
        const f_plan pl = f_plan_for(true, Nreg, U, GW, ctx->knobs.f_form);
        const size_t need = (size_t)GW * Nreg * Nreg * 64;
        if ((pl.form == F_PAIR || pl.form == F_PAIRX) && need <= ((size_t)8 << 30)) {
            int rc = fcd_fsq_reserve(ctx, need);
            if (rc) return rc;
            fsq = (uint8_t *)ctx->fsq;
        }
    }
    for (int64_t i = 0; i < n_sweeps; ++i) {
        int rc = fcd_gibbs_f_step_sq(ctx, S_B, lM, lMf, hyper, f_state, r_bits, Nreg, U, G, chain0, seed, sweep0 + i,
                                     (hipStream_t)stream, fsq, false);
        if (rc) return rc;
        rc = fcd_gibbs_r_step_sq(ctx, lM, lMd, hyper, f_state, r_bits, Nreg, U, G, chain0, seed, sweep0 + i, edge_mode,
                                 (hipStream_t)stream, fsq);
        if (rc) return rc;
    }
    if (counts && n_sweeps > 0) return fcd_gibbs_stats(ctx, f_state, r_bits, Nreg, U, G, counts, stream);
    return FCD_OK;
}

extern "C" int fcd_gibbs_mstep(fcd_ctx *ctx, const int64_t *counts, int64_t Nreg, int64_t U, double *hyper,
                               fcd_stream stream) {
    if (!ctx || !counts || !hyper) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_mstep: null pointer");
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    hipLaunchKernelGGL(gibbs_mstep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long *>(counts), (double)(Nreg * U), (double)fcd_tri(Nreg), hyper);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_accumulate(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg,
                                    int64_t U, int64_t G, uint32_t *cnt_f, uint32_t *cnt_r, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!f_state || !r_bits || !cnt_f || !cnt_r) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_accumulate: null pointer");
    int64_t blocks = (g.C + 3) / 4;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(gibbs_accum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f_state, r_bits, g.C,
                       Nreg * U, g.GW, G, cnt_f, cnt_r);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

// one launch of gibbs_tally_kernel; counts / cnt_f+cnt_r / hyper / r_U each optional
static int launch_tally(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                        const fcd_geo &g, int64_t *counts, uint32_t *cnt_f, uint32_t *cnt_r, double *hyper, uint32_t *r_U,
                        int ru_words, hipStream_t s, bool f_done = false) {
    tally_args a;
    a.f_state = f_state; a.r_bits = r_bits;
    a.C = g.C; a.NU = Nreg * U; a.G = G;
    a.GW = g.GW; a.Nreg = (int)Nreg; a.U = (int)U; a.NW = ru_words;
    a.acc = (counts || hyper) ? (unsigned long long *)ctx->acc : nullptr;
    a.counts_out = reinterpret_cast<unsigned long long *>(counts);
    a.cnt_f = cnt_f; a.cnt_r = cnt_r;
    a.hyper = hyper; a.r_U = r_U;
    int64_t blocks = (g.C + 63) / 64;          // 16 waves x 4 edges per workgroup and round
    const int64_t cap = (int64_t)ctx->num_cu * 2;      // (8 per CU measured slower: 19.6 us against 15.6 us at cfg3)
    if (blocks > cap) blocks = cap;
    a.n_f_blocks = f_done ? 0 : (int)blocks;
    {   // the r bits alone: one thread per (region, patient), a few blocks
        int64_t rb = (a.NU + 1023) / 1024;
        if (rb > 64) rb = 64;
        a.n_r_blocks = (int)rb;
        if (f_done) blocks = rb;
    }
    if (r_U) {
        int64_t ru_blocks = ((int64_t)g.GW * Nreg * a.NW + 15) / 16;
        if (ru_blocks > ctx->num_cu) ru_blocks = ctx->num_cu;
        blocks += ru_blocks;
    }
    hipLaunchKernelGGL(gibbs_tally_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, a);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_tally(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                               int64_t G, int64_t *counts, uint32_t *cnt_f, uint32_t *cnt_r, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_tally: null pointer");
    if ((cnt_f == nullptr) != (cnt_r == nullptr)) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_tally: cnt_f and cnt_r go together");
    if (!counts && !cnt_f) return FCD_OK;
    return launch_tally(ctx, f_state, r_bits, Nreg, U, G, g, counts, cnt_f, cnt_r, nullptr, nullptr, 0, (hipStream_t)stream);
}

// The sampler loop of ONE rank between two exchanges of pooled statistics (what UnsharedRegionFit(method='gibbs'),
// run_chains and bench.py call).  Per sweep: f pass (1 launch), packing for the r pass (1), block steps of the r pass
// (ceil(Nreg/16) + 1), tally (1) -- the tally also carries the M-step and the slot words of the next f pass.
extern "C" int fcd_gibbs_run(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *lMd,
                             double *hyper, uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                             int64_t chain0, uint64_t seed, int64_t sweep0, int64_t n_sweeps, int edge_mode,
                             int64_t mstep_every, int64_t accumulate_from, int64_t *counts, uint32_t *cnt_f, uint32_t *cnt_r,
                             fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, chain0, g);
    if (rc) return rc;
    if (n_sweeps < 0 || sweep0 < 0 || sweep0 + n_sweeps > (1ll << 32))
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_run: sweep range [%lld, +%lld) outside the 32-bit counter word", sweep0, n_sweeps);
    if (!S_B || !lM || !hyper || !f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_run: null pointer");
    if ((cnt_f == nullptr) != (cnt_r == nullptr)) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_run: cnt_f and cnt_r go together");
    if (mstep_every < 0) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_run: mstep_every < 0");
    hipStream_t s = (hipStream_t)stream;
    // square copy of the f state for the r pass's packing (see fcd_gibbs_sweeps)
    uint8_t *fsq = nullptr;
    const f_plan pl = f_plan_for(lMf != nullptr, Nreg, U, g.GW, ctx->knobs.f_form);
    const bool pair_form = pl.form == F_PAIR || pl.form == F_PAIRX;
    if (pair_form && lMd && edge_mode == FCD_EDGE_SYMMETRIC && true) {
        const size_t need = (size_t)g.GW * Nreg * Nreg * 64;
        if (need <= ((size_t)8 << 30)) {
            rc = fcd_fsq_reserve(ctx, need);
            if (rc) return rc;
            fsq = (uint8_t *)ctx->fsq;
        }
    }
    // The slot words of the f pass live BEHIND the r pass's workspace inside this loop (on their own they sit at its head):
    // the r pass's two buffers of panel values then survive from one sweep to the next, and since a completed pipelined
    // pass leaves every slot holding its sentinel again, only the first sweep of a call has to write them (13 MB at cfg3).
    size_t ru_off = 0;
    if (pair_form) {
        ru_off = (fcd_r_pass_ws_bytes(Nreg, U, g.GW, ctx->knobs.r_path) + 511) / 512 * 512;
        rc = fcd_ws_reserve(ctx, ru_off + fcd_f_pass_ws_bytes(Nreg, U, g.GW));      // the tally writes the next pass's slot words there
        if (rc) return rc;
    }
    bool ru_ready = false;
    bool sentinels_in_place = false;
    for (int64_t i = 0; i < n_sweeps; ++i) {
        rc = fcd_gibbs_f_step_sq(ctx, S_B, lM, lMf, hyper, f_state, r_bits, Nreg, U, G, chain0, seed, sweep0 + i, s, fsq, ru_ready, ru_off);
        if (rc) return rc;
        const bool last = i + 1 == n_sweeps;
        const bool do_m = mstep_every > 0 && (i + 1) % mstep_every == 0;
        const bool do_a = cnt_f && sweep0 + i >= accumulate_from;
        // the f half of this sweep's tally rides in the r pass's packing launch (beside it, not after the pass)
        fcd_tally_f tf;
        tf.f_state = f_state; tf.C = g.C; tf.G = G; tf.GW = g.GW;
        tf.acc = (do_m || (last && counts)) ? (unsigned long long *)ctx->acc : nullptr;
        tf.cnt_f = do_a ? cnt_f : nullptr;
        const bool want_f = tf.acc || tf.cnt_f;
        bool f_done = false;
        rc = fcd_gibbs_r_step_sq(ctx, lM, lMd, hyper, f_state, r_bits, Nreg, U, G, chain0, seed, sweep0 + i, edge_mode, s, fsq,
                                 want_f ? &tf : nullptr, &f_done, sentinels_in_place);
        if (rc) {
            // the packing launch may already have added this sweep's f counts into the context's accumulators and no tally
            // will draw the ticket that zeroes them: leave them clean for the next call
            if (tf.acc) (void)hipMemsetAsync(ctx->acc, 0, 8 * sizeof(unsigned long long), s);
            return rc;
        }
        sentinels_in_place = pair_form && ctx->r_form_last == 2;        // (a pipelined pass has just been queued)
        // the r pass's scratch is dead once its last launch is queued: the slot words of the next f pass go to its place
        uint32_t *r_U_next = (pair_form && !last) ? (uint32_t *)((char *)ctx->ws + ru_off) : nullptr;
        int64_t *cts = (last ? counts : nullptr);
        // Several ranks (a communicator on the context): the tally leaves this rank's counts in the context's vector, RCCL sums
        // it over the ranks in place ON THIS STREAM, a one-thread kernel makes the M-step from the pooled counts -- all queued
        // behind one another, the host far ahead.  One rank: the M-step runs inside the tally launch.
        const bool pooled = do_m && ctx->comm != nullptr;
        int64_t *tally_counts = pooled ? (int64_t *)ctx->pool_counts : cts;
        if (do_m || do_a || cts || r_U_next) {
            rc = launch_tally(ctx, f_state, r_bits, Nreg, U, G, g, tally_counts, do_a ? cnt_f : nullptr, do_a ? cnt_r : nullptr,
                              (do_m && !pooled) ? hyper : nullptr, r_U_next, pl.NW, s, f_done);
            if (rc) {
                if (tf.acc) (void)hipMemsetAsync(ctx->acc, 0, 8 * sizeof(unsigned long long), s);
                return rc;
            }
        }
        if (pooled) {
            rc = fcd_comm_allreduce_counts(ctx, (long long *)ctx->pool_counts, s);
            if (rc) return rc;
            hipLaunchKernelGGL(gibbs_mstep_kernel, dim3(1), dim3(1), 0, s, (const long long *)ctx->pool_counts, (double)(Nreg * U),
                               (double)fcd_tri(Nreg), hyper);
            FCD_LAUNCH_CHECK();
            if (cts) FCD_HIP_TRY(hipMemcpyAsync(cts, ctx->pool_counts, 8 * sizeof(long long), hipMemcpyDeviceToDevice, s));
        } else if (cts && ctx->comm) {
            // (counts asked for without an M-step in this sweep: pooled all the same -- what a caller of several ranks expects)
            rc = fcd_comm_allreduce_counts(ctx, (long long *)cts, s);
            if (rc) return rc;
        }
        ru_ready = r_U_next != nullptr;
    }
    return FCD_OK;
}

extern "C" int fcd_gibbs_logjoint(fcd_ctx *ctx, const double *S_B, const double *lM, const double *hyper,
                                  const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                                  double *out, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!S_B || !lM || !hyper || !f_state || !r_bits || !out) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_logjoint: null pointer");
    const size_t need = (size_t)LJ_SLICES * g.GW * 64 * sizeof(double);
    rc = fcd_ws_reserve(ctx, need);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gibbs_logjoint_kernel, dim3((unsigned)g.GW, LJ_SLICES), dim3(64), 0, s, S_B, lM, hyper, f_state,
                       r_bits, (int)Nreg, (int)U, g.C, g.GW, (double *)ctx->ws);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gibbs_logjoint_fold, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, (const double *)ctx->ws,
                       (int64_t)g.GW * 64, G, out);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_chain_rsum(fcd_ctx *ctx, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, uint32_t *out,
                                    fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!r_bits || !out) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_chain_rsum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    FCD_HIP_TRY(hipMemsetAsync(out, 0, (size_t)G * sizeof(uint32_t), s));
    const int64_t NU = Nreg * U;
    const int slices = (int)(NU < 64 ? NU : 64);
    hipLaunchKernelGGL(gibbs_rsum_kernel, dim3((unsigned)g.GW, (unsigned)slices), dim3(64), 0, s, r_bits, NU, G, out);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_conditionals(fcd_ctx *ctx, const double *S_B, const double *lM, const double *hyper,
                                      const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                                      int64_t G, int edge_mode, double *cond_f, double *cond_r, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!S_B || !lM || !hyper || !f_state || !r_bits) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_conditionals: null pointer");
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_conditionals: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2 && cond_r)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    hipStream_t s = (hipStream_t)stream;
    if (cond_f) {
        rc = launch_f<true>(ctx, S_B, lM, hyper, const_cast<uint8_t *>(f_state), r_bits, Nreg, U, G, g, 0, 0, 0, cond_f, s);
        if (rc) return rc;
    }
    if (cond_r) {
        if (U > 65535 || Nreg > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_gibbs_conditionals: grid too large");
        hipLaunchKernelGGL(gibbs_cond_r_kernel, dim3((unsigned)U, (unsigned)g.GW, (unsigned)Nreg), dim3(64), 0, s, lM, hyper,
                           f_state, r_bits, (int)Nreg, (int)U, g.C, G, edge_mode, cond_r);
        FCD_LAUNCH_CHECK();
    }
    return FCD_OK;
}

extern "C" int fcd_gibbs_export_state(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg,
                                      int64_t U, int64_t G, uint8_t *f, uint8_t *r, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!f_state || !r_bits || !f || !r) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_export_state: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(export_f_kernel, dim3((unsigned)((G * g.C + 255) / 256)), dim3(256), 0, s, f_state, g.C, G, f);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(export_r_kernel, dim3((unsigned)((G * Nreg * U + 255) / 256)), dim3(256), 0, s, r_bits, Nreg * U, G, r);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_gibbs_import_state(fcd_ctx *ctx, const uint8_t *f, const uint8_t *r, int64_t Nreg, int64_t U,
                                      int64_t G, uint8_t *f_state, uint64_t *r_bits, fcd_stream stream) {
    fcd_geo g;
    int rc = fcd_geo_check(ctx, Nreg, U, G, 0, g);
    if (rc) return rc;
    if (!f_state || !r_bits || !f || !r) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_gibbs_import_state: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nf = (int64_t)g.GW * g.C * 64;
    hipLaunchKernelGGL(import_f_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, f, g.C, G, g.GW, f_state);
    FCD_LAUNCH_CHECK();
    const int64_t words = (int64_t)g.GW * Nreg * U;
    hipLaunchKernelGGL(import_r_kernel, dim3((unsigned)((words + 3) / 4)), dim3(256), 0, s, r, Nreg * U, G, g.GW, r_bits);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_philox_uniforms(fcd_ctx *ctx, const uint32_t *ctr4, int64_t n, uint64_t seed, double *out,
                                   fcd_stream stream) {
    if (!ctx || !ctr4 || !out || n < 0) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_philox_uniforms: bad argument");
    if (n == 0) return FCD_OK;
    hipLaunchKernelGGL(philox_uniforms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ctr4, n,
                       seed, out);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
